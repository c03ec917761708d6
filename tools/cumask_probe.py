#!/usr/bin/env python3
"""(GPU box) The three lanes on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) against three lanes sharing all 256:
does a lane's kernel mix better with the other lanes' kernels, or with more of its own wavefronts?  bench.py's timed
configuration (staged form, 256-frame batches rotating over five).  usage: cumask_probe.py [steps]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_uwspr_amd as G  # noqa: E402
from gr_uwspr_amd import dist as D  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, NB, NL = 256, 5, 3
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
N = G.native
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 104729 * k, snr_db=-20.0) for k in range(NB)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(name, streams):
    for s in streams:
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    lanes = []
    for k in range(len(streams)):
        cx = G.Context(device=0, options={"sched": 0})
        cx.set_stream(streams[k].cuda_stream)
        lanes.append({"stream": streams[k], "ctx": cx,
                      "cands": torch.empty(B * cx.maxfreqs * 48, dtype=torch.uint8, device=dev),
                      "npk": torch.empty(B, dtype=torch.int32, device=dev),
                      "out": torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                      "slab": torch.zeros((B, D.SLAB_BYTES), dtype=torch.uint8, device=dev)})

    def step(i):
        ln = lanes[i % len(lanes)]
        with torch.cuda.stream(ln["stream"]):
            ln["ctx"].pipeline_slabs(D.SLAB_K, ln["slab"])
            ln["ctx"].pipeline_batch_into(batches[i % NB], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)

    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print("%-64s %.4f ms per step  %8.0f frames/s" % (name, 1e3 * t / K, B * K / t), flush=True)
    torch.cuda.synchronize()
    for ln in lanes:
        ln["ctx"].close()


ncu = torch.cuda.get_device_properties(0).multi_processor_count
run("three lanes, plain streams (all %d CUs each)" % ncu, [torch.cuda.Stream(device=dev) for _ in range(NL)])
run("three lanes, full masks (control: masked streams, all CUs)", [masked_stream(range(ncu)) for _ in range(NL)])
third = [range(0, 86), range(86, 171), range(171, 256)]
run("three lanes, mask bits in three contiguous thirds", [masked_stream(t) for t in third])
run("three lanes, mask bits interleaved (bit mod 3)", [masked_stream([b for b in range(ncu) if b % 3 == k]) for k in range(NL)])
run("six lanes, two per third", [masked_stream(third[k % 3]) for k in range(6)])
run("three lanes, plain streams again", [torch.cuda.Stream(device=dev) for _ in range(NL)])
