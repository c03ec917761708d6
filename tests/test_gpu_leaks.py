"""Nothing accumulates over open / close cycles: device memory, host memory, threads.

Contexts (uwspr_ctx_create builds tables, scratch grows on demand and is owned by the context), the stream ring,
the staged form's per-slot tables, and the pipe (its lanes' contexts, device ring, page-locked staging and result
buffers, coordinator threads on the process-wide pool)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rss_kb():
    return int(open("/proc/self/statm").read().split()[1]) * 4


def test_a_thousand_context_cycles_leak_nothing(G):
    import torch
    fr = G.synth.make_frames(16, seed=3, snr_db=-18.0)
    free0 = rss0 = None
    for it in range(1000):
        c = G.Context(halfbandwidth=10 + (it % 3) * 10, maxdrift=it % 2, options={"sched": it % 2})
        if it % 8 == 0:                                   # every eighth one does real work (scratch buffers grow)
            c.pipeline_batch(fr, max_per_frame=2)
            c.frontend(np.zeros((1, 1000), np.float32))
        c.close()
        if it == 40:
            torch.cuda.synchronize()
            free0, rss0 = torch.cuda.mem_get_info()[0], _rss_kb()
    torch.cuda.synchronize()
    free1, rss1 = torch.cuda.mem_get_info()[0], _rss_kb()
    assert free0 - free1 < (64 << 20), "device memory: %d KB lost over 960 cycles" % ((free0 - free1) >> 10)
    assert rss1 - rss0 < (256 << 10), "host memory: %d KB gained over 960 cycles" % (rss1 - rss0)


def test_pipe_and_stream_ring_cycles_leak_nothing(G):
    import torch
    fr = G.synth.make_frames(12, seed=5, snr_db=-18.0)
    stream = np.concatenate([fr[k][:10 * 3375] for k in range(12)])
    free0 = None
    for it in range(40):
        c = G.Context(options={"sched": 0})
        c.stream_open(3375, 4)
        c.stream_push(stream[:60000])
        ptr, stride, pos = c.stream_take_view(4)
        c.set_frame_stride(stride)
        c.pipeline_batch(G.FrameView(4, ptr=ptr), max_per_frame=1)
        c.close()
        p = G.Pipe(hop=3375, batch_frames=8, max_per_frame=1, lanes=3)
        p.push(stream)
        p.flush()
        n = len(p.collect())
        p.close()
        assert n > 0
        if it == 5:
            torch.cuda.synchronize()
            free0, rss0, thr0 = torch.cuda.mem_get_info()[0], _rss_kb(), threading.active_count()
    torch.cuda.synchronize()
    free1, rss1 = torch.cuda.mem_get_info()[0], _rss_kb()
    assert free0 - free1 < (64 << 20), "device memory: %d KB lost over 34 cycles" % ((free0 - free1) >> 10)
    assert rss1 - rss0 < (256 << 10), "host memory: %d KB gained over 34 cycles" % (rss1 - rss0)
    assert threading.active_count() <= thr0
