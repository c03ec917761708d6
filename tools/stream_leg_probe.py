#!/usr/bin/env python3
"""The quiet-stream leg of bench.py alone (uwspr_pipe_acquire/commit, PCIe-inclusive), one fresh process per
environment setting:  python tools/stream_leg_probe.py ["ENV=a ENV2=b" ...]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for setting in sys.argv[1:]:
        env = dict(os.environ)
        for kv in setting.split():
            k, v = kv.split("=")
            env[k] = v
        for rep in range(2):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print("%-40s %s" % (setting, r.stdout.strip().split("\n")[-1]), flush=True)
    sys.exit(0)
import torch
import gr_uwspr_amd as G
B, hop, KS = 256, 3375, 100
pipe = G.Pipe(hop=hop, batch_frames=B, max_per_frame=1, lanes=3)
rng = np.random.default_rng(3)
sig = G.synth.make_frames(20, seed=99, snr_db=-20.0)[:, 375:375 + 162 * 256]   # with its own noise, as bench.py builds it
for k in range(4):
    buf = pipe.acquire(B * hop)
    buf[:] = (G.synth.sigma_for_snr(-20.0) * rng.standard_normal((B * hop, 2))).astype(np.float32)
    if os.environ.get("PROBE_SIGNALS", "1") != "0":
        for t in range(0, B * hop // 45000 - 1, 40):
            s0 = t * 45000 + int(rng.integers(0, 3000))
            buf[s0:s0 + sig.shape[1]] += sig[t % 20]
    pipe.commit(B * hop)
pipe.flush(); pipe.collect()
rates = []
for _ in range(5):
    t0 = time.perf_counter(); f0 = pipe.stats()["frames"]
    for i in range(KS):
        pipe.acquire(B * hop); pipe.commit(B * hop)
        if i % 8 == 7:
            pipe.collect()
    pipe.flush(); pipe.collect()
    rates.append((pipe.stats()["frames"] - f0) / (time.perf_counter() - t0))
st = pipe.stats()
print("quiet stream %.0f frames/s (min %.0f max %.0f) frames %d decoded %d timeouts %d calls %d resumed %d gpu_wait %.2f fano %.2f resume %.2f" % (float(np.median(rates)), min(rates), max(rates), st["frames"], st["decoded"], st["fano_timeouts"], st["fano_calls"], st["resumed"], st["gpu_wait_s"], st["fano_s"], st["resume_s"]))
pipe.close()
