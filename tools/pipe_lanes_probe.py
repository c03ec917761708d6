#!/usr/bin/env python3
"""End-to-end decoded frames/s of uwspr_pipe_* (frames resident in HBM) against the number of lanes and the
schedule form.  GPU box:  python tools/pipe_lanes_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_uwspr_amd as G

B, NB, KS = 256, 5, 100
dev = torch.device("cuda", 0)
batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 104729 * k, snr_db=-20.0) for k in range(NB)]
torch.cuda.synchronize()
pipes = {(f, l): G.Pipe(batch_frames=B, max_per_frame=1, lanes=l, sched=f) for f in ("staged", "fused") for l in (2, 3, 4)}
for (f, l), pipe in pipes.items():
    for i in range(3 * NB):
        pipe.submit_device(batches[i % NB], B)
    pipe.flush(); pipe.collect()
    rates = []
    for _ in range(4):
        t0 = time.perf_counter()
        for i in range(KS):
            pipe.submit_device(batches[i % NB], B)
            if i % 8 == 7:
                pipe.collect()
        pipe.flush(); pipe.collect()
        rates.append(KS * B / (time.perf_counter() - t0))
    st = pipe.stats()
    print("%-6s lanes %d: %8.0f frames/s (min %8.0f max %8.0f) decoded %.4f" %
          (f, l, float(np.median(rates)), min(rates), max(rates), st["decoded"] / max(st["candidates"], 1)), flush=True)
    pipe.close()
