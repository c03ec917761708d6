#!/usr/bin/env python3
"""(GPU box) Does a hipGraph of the step help?  The timed configuration of bench.py (staged form, three lanes, 256-frame
batches rotating over five) with every (lane, batch) step captured ONCE into a graph (stream capture of the library's
own launches through torch.cuda.graph) and replayed, against the same steps launched eagerly.  usage: graph_probe.py [steps] [lanes]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_uwspr_amd as G  # noqa: E402
from gr_uwspr_amd import dist as D  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, NB = 256, 5
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
N = G.native
streams = [torch.cuda.Stream(device=dev) for _ in range(NL)]
for s in streams:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 104729 * k, snr_db=-20.0) for k in range(NB)]
lanes = []
for k in range(NL):
    cx = G.Context(device=0, options={"sched": 0})
    cx.set_stream(streams[k].cuda_stream)
    lanes.append({"stream": streams[k], "ctx": cx,
                  "cands": torch.empty(B * cx.maxfreqs * 48, dtype=torch.uint8, device=dev),
                  "npk": torch.empty(B, dtype=torch.int32, device=dev),
                  "out": torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                  "slab": torch.zeros((B, D.SLAB_BYTES), dtype=torch.uint8, device=dev)})


def step(i):
    ln = lanes[i % NL]
    with torch.cuda.stream(ln["stream"]):
        ln["ctx"].pipeline_slabs(D.SLAB_K, ln["slab"])
        ln["ctx"].pipeline_batch_into(batches[i % NB], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)


def region(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, t1 - t0


for i in range(30):
    step(i)
torch.cuda.synchronize()
ref = [ln["out"].clone() for ln in lanes]
eager = sorted(region(step, K) for _ in range(3))[1]
print("eager : %.4f ms per step (%.0f frames/s), host enqueue %.4f ms per step" % (1e3 * eager[0] / K, B * K / eager[0], 1e3 * eager[1] / K))
graphs = {}
try:
    for l in range(NL):
        for b in range(NB):
            g = torch.cuda.CUDAGraph()
            ln = lanes[l]
            with torch.cuda.graph(g, stream=ln["stream"]):
                ln["ctx"].pipeline_slabs(D.SLAB_K, ln["slab"])
                ln["ctx"].pipeline_batch_into(batches[b], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)
            graphs[(l, b)] = g
except Exception as e:                         # noqa: BLE001
    print("capture failed:", repr(e)[:300])
    sys.exit(0)


def gstep(i):
    with torch.cuda.stream(lanes[i % NL]["stream"]):
        graphs[(i % NL, i % NB)].replay()


for i in range(30):
    gstep(i)
torch.cuda.synchronize()
gr = sorted(region(gstep, K) for _ in range(3))[1]
print("graphs: %.4f ms per step (%.0f frames/s), host enqueue %.4f ms per step" % (1e3 * gr[0] / K, B * K / gr[0], 1e3 * gr[1] / K))
# the same bytes?
for i in range(NL * NB):
    gstep(i)
torch.cuda.synchronize()
a = [ln["out"].clone() for ln in lanes]
for i in range(NL * NB):
    step(i)
torch.cuda.synchronize()
print("graph replay gives the eager bytes:", all(torch.equal(x, ln["out"]) for x, ln in zip(a, lanes)))
eager2 = sorted(region(step, K) for _ in range(3))[1]
print("eager : %.4f ms per step (%.0f frames/s) again" % (1e3 * eager2[0] / K, B * K / eager2[0]))
