#!/usr/bin/env python3
"""The busy-stream leg of bench.py alone (a transmission in every window; host-bound by Fano time-outs) for several
numbers of pipe lanes, one fresh process each:  python tools/busy_lanes_probe.py 3 6 8"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lanes in sys.argv[1:]:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lanes], capture_output=True, text=True)
        print("lanes %-3s %s" % (lanes, (r.stdout.strip().split("\n") or [r.stderr[-300:]])[-1]), flush=True)
    sys.exit(0)
import gr_uwspr_amd as G
lanes = int(sys.argv[2])
B, hop = 256, 3375
pipe = G.Pipe(hop=hop, batch_frames=B, max_per_frame=1, lanes=lanes)
rng = np.random.default_rng(3)
sig = G.synth.make_frames(20, seed=99, snr_db=-20.0)[:, 375:375 + 162 * 256]
for k in range(4):
    buf = pipe.acquire(B * hop)
    buf[:] = (G.synth.sigma_for_snr(-20.0) * rng.standard_normal((B * hop, 2))).astype(np.float32)
    for t in range(0, B * hop // 45000 - 1, 1):
        s0 = t * 45000 + int(rng.integers(0, 3000))
        buf[s0:s0 + sig.shape[1]] += sig[t % 20]
    pipe.commit(B * hop)
pipe.flush(); pipe.collect()
rates = []
for _ in range(3):
    t0 = time.perf_counter(); f0 = pipe.stats()["frames"]
    for i in range(16):
        pipe.acquire(B * hop); pipe.commit(B * hop)
        if i % 8 == 7:
            pipe.collect()
    pipe.flush(); pipe.collect()
    rates.append((pipe.stats()["frames"] - f0) / (time.perf_counter() - t0))
st = pipe.stats()
print("busy stream %.0f frames/s (min %.0f max %.0f) timeouts/frame %.4f gpu_wait %.2f fano %.2f resume %.2f" % (
    float(np.median(rates)), min(rates), max(rates), st["fano_timeouts"] / max(st["frames"], 1), st["gpu_wait_s"], st["fano_s"], st["resume_s"]))
pipe.close()
