#!/usr/bin/env python3
"""End-to-end decoded frames/s of uwspr_pipe_* (frames resident in HBM) against the number of lanes and the
schedule form -- one fresh process per configuration (HIP maps streams onto a few hardware queues in creation
order; streams of other pipes in the same process would share them).
GPU box:  python tools/pipe_lanes_probe.py            (the sweep)
          python tools/pipe_lanes_probe.py staged 4   (one configuration)"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) < 3:
    for q in (None, "8"):
        for f in ("staged", "fused"):
            for l in (2, 3, 4, 6):
                env = dict(os.environ)
                if q:
                    env["GPU_MAX_HW_QUEUES"] = q
                r = subprocess.run([sys.executable, os.path.abspath(__file__), f, str(l)], env=env, capture_output=True, text=True)
                print("GPU_MAX_HW_QUEUES=%-7s %s" % (q or "default", r.stdout.strip().split("\n")[-1]), flush=True)
    sys.exit(0)

import torch
import gr_uwspr_amd as G
f, l = sys.argv[1], int(sys.argv[2])
B, NB, KS = 256, 5, 100
dev = torch.device("cuda", 0)
pipe = G.Pipe(batch_frames=B, max_per_frame=1, lanes=l, sched=f)
batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 104729 * k, snr_db=-20.0) for k in range(NB)]
torch.cuda.synchronize()
for i in range(3 * NB):
    pipe.submit_device(batches[i % NB], B)
pipe.flush(); pipe.collect()
rates = []
for _ in range(5):
    t0 = time.perf_counter()
    for i in range(KS):
        pipe.submit_device(batches[i % NB], B)
        if i % 8 == 7:
            pipe.collect()
    pipe.flush(); pipe.collect()
    rates.append(KS * B / (time.perf_counter() - t0))
st = pipe.stats()
print("%-6s lanes %d: %8.0f frames/s (min %8.0f max %8.0f) decoded %.4f" %
      (f, l, float(np.median(rates)), min(rates), max(rates), st["decoded"] / max(st["candidates"], 1)))
pipe.close()
