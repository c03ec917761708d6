#!/usr/bin/env python3
"""K0 front-end alone: frames/s and the FP32-FMA rate of both tap sets (audio resident in HBM).
    python tools/k0_time.py [B]     -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gr_uwspr_amd as G

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
audio = torch.randn((B, 45000 * 32), device=dev, dtype=torch.float32)
res = {"frames": B, "nin": 45000 * 32}
for mode, name in ((0, "grc"), (1, "compact")):
    ctx = G.Context(options={"frontend": mode})
    g, _ = G.frontend_design(mode, 0)
    J = (len(g) + 31) // 32
    J = (J + 7) // 8 * 8
    ctx.frontend(audio)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.frontend(audio)
        ctx.synchronize()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 4)
    dt = float(np.median(ts))
    flop = 2.0 * 2.0 * 32 * J * 45000 * B          # complex tap x real sample, fused multiply-adds, padded taps
    res[name] = {"taps": len(g), "taps_padded": 32 * J, "ms": 1e3 * dt, "frames_per_s": B / dt,
                 "fp32_tflops": flop / dt / 1e12, "frac_of_157.3": flop / dt / 1e12 / 157.3,
                 "input_GBs": B * 45000 * 32 * 4 / dt / 1e9}
    ctx.close()
print(json.dumps(res))
