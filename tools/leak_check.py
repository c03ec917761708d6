import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import gr_uwspr_amd as G
fr = G.synth.make_frames(16, seed=3, snr_db=-18.0)
free0 = None
for it in range(120):
    c = G.Context(halfbandwidth=10 + (it % 3) * 10, maxdrift=it % 2)
    cands, out = c.pipeline_batch(fr, max_per_frame=2)
    a = c.frontend(np.zeros((1, 1000), np.float32))
    c.close()
    if it == 10:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print("free after 10 iterations %d MB, after 120 iterations %d MB, delta %d KB" % (free0 >> 20, free1 >> 20, (free0 - free1) >> 10))
assert free0 - free1 < 64 << 20
print("leak check ok")

# round 3: the same for the stream ring, the staged form's per-slot tables and the pipe (its lanes' contexts, the
# device ring, the page-locked staging buffers and result buffers, its coordinator threads)
import os, threading
os.environ["UWSPR_OPTIONS"] = "sched=0"
stream = np.concatenate([fr[k][:10 * 3375] for k in range(12)])
free0 = None
for it in range(40):
    c = G.Context()
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
    if it == 5:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; thr0 = threading.active_count()
        rss0 = int(open("/proc/self/statm").read().split()[1])
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
rss1 = int(open("/proc/self/statm").read().split()[1])
print("pipe/stream: %d records per pass; device delta %d KB, host RSS delta %d KB over 34 open/close cycles"
      % (n, (free0 - free1) >> 10, (rss1 - rss0) * 4))
assert free0 - free1 < 64 << 20 and (rss1 - rss0) * 4 < 256 << 10
print("leak check ok (pipe, stream ring)")
