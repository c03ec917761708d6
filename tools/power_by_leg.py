#!/usr/bin/env python3
"""(GPU box) Socket power and shader clock (rocm-smi, sampled in a thread) while ONE part of the path runs in a loop on
one stream: the coarse search alone (K1-K3), the staged and the fused schedule alone (K4 + folds / K6 on given candidates),
the configs[2] grid sweep (K4 at its best: 0.68 of the no-FMA peak).  With the ~490 W of base power from
profiles/r05_clock_trace.txt this gives the energy per vector lane-instruction of each kernel family.
    python3 tools/power_by_leg.py [seconds per leg]  -> stdout"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402
from gr_uwspr_amd import sweep as SW  # noqa: E402

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples = []
stop = False


def sampler():
    while not stop:
        try:
            c = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, timeout=5).stdout)
            p = json.loads(subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, timeout=5).stdout)
            k = sorted(c)[0]
            sclk = next(int(re.search(r"\((\d+)", v).group(1)) for n, v in c[k].items() if "sclk" in n.lower() and "(" in str(v))
            w = next(float(v) for n, v in p[k].items() if "power" in n.lower())
            samples.append((time.time(), sclk, w))
        except Exception:          # noqa: BLE001 -- a sample lost
            pass


def leg(name, fn, units, unit_name):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < SEC:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    t1 = time.time()
    s = [(c, w) for t, c, w in samples if t0 + 1.0 <= t <= t1 - 0.2]
    clk = sorted(c for c, _ in s); pw = sorted(w for _, w in s)
    rate = n * units / (t1 - t0)
    print("%-34s %9.0f %s/s   sclk %4d MHz   power %6.0f W  (%d samples)   %.3f mJ per %s above 490 W"
          % (name, rate, unit_name, clk[len(clk) // 2] if clk else 0, pw[len(pw) // 2] if pw else 0, len(s),
             1e3 * ((pw[len(pw) // 2] if pw else 0) - 490.0) / rate, {"frames": "frame", "hypotheses": "hypothesis"}.get(unit_name, unit_name)))


def main():
    global stop
    th = threading.Thread(target=sampler, daemon=True); th.start()
    dev = torch.device("cuda", 0)
    B = 256
    fr = G.synth.make_frames_torch(B, dev, seed=0xC0FFEE, snr_db=-20.0)
    N = G.native
    for sched, name in ((0, "staged"), (1, "fused")):
        c = G.Context(options={"sched": sched})
        cands = torch.empty(B * c.maxfreqs * 48, dtype=torch.uint8, device=dev)
        npk = torch.empty(B, dtype=torch.int32, device=dev)
        out = torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        leg("whole path, %s, one stream" % name, lambda: c.pipeline_batch_into(fr, cands, npk, out, max_per_frame=1), B, "frames")
        c.close()
    c = G.Context()
    leg("coarse search alone (K1-K3)", lambda: c.fdr_batch_device(fr) if hasattr(c, "fdr_batch_device") else c.fdr_batch(fr), B, "frames")
    Bs = 1024
    fr2 = G.synth.make_frames_torch(Bs, dev, seed=0x5EED, snr_db=-20.0)
    H = Bs * 200
    cent = np.zeros(Bs, N.CAND_DTYPE); cent["freq"] = 0.0; cent["shift"] = 368
    cent_t = torch.from_numpy(np.frombuffer(cent.tobytes(), np.uint8).copy()).to(dev)
    df = np.array(SW.DF_STEPS, np.float32) * np.float32(0.25); dd = np.array(SW.DRIFTS, np.float32); dl = np.array(SW.LAGS, np.int32)
    sync_t = torch.empty(H, dtype=torch.float32, device=dev); sym_t = torch.empty(H * 162, dtype=torch.uint8, device=dev)
    leg("configs[2] grid sweep (K4 grid + K5)", lambda: c.sync_grid(fr2, cent_t, df, dd, dl, into=(sync_t, sym_t)), H, "hypotheses")
    c.close()
    stop = True


if __name__ == "__main__":
    main()
