#!/usr/bin/env python3
"""Soak: whole pipeline (FDR + schedule) on many frames at mixed SNRs, GPU vs the CPU
oracle, every candidate of every frame.  usage: soak_parity.py [nframes] [halfbandwidth] [maxdrift] [seed base, default 5000]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gr_uwspr_amd as G  # noqa: E402
import oracle_py as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    hbw = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    maxdrift = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
    snrs = [-32.0, -28.0, -26.0, -24.0, -20.0, -10.0, None]
    parts = []
    for k, snr in enumerate(snrs):
        m = (n + len(snrs) - 1) // len(snrs)
        fr = G.synth.make_frames(m, seed=seed0 + 100 * k, snr_db=snr, halfbandwidth=hbw, maxdrift=float(maxdrift))
        if snr == -32.0:
            fr[: m // 2] = (np.random.default_rng(9 + seed0 - 5000).standard_normal(fr[: m // 2].shape) * 0.5).astype(np.float32)
        parts.append(fr)
    frames = np.concatenate(parts)[:n]
    per = 4
    ctx = G.Context(halfbandwidth=hbw, maxdrift=maxdrift)
    print("options: %s -> sched %d, stage_kernels %d, k4_forms %d, reuse %d"
          % (os.environ.get("UWSPR_OPTIONS") or "(default)", ctx.get_option("sched"), ctx.get_option("stage_kernels"),
             ctx.get_option("k4_forms"), ctx.get_option("reuse")))
    t0 = time.time()
    cands, out = ctx.pipeline_batch(frames, max_per_frame=per)
    tg = time.time() - t0
    O.lib(); O.pr3()
    nw = min(16, len(os.sched_getaffinity(0)))
    fdrs = [O.FDR(halfbandwidth=hbw, maxdrift=maxdrift) for _ in range(nw)]
    bad = []

    def check(args):
        w, b = args
        exp = fdrs[w].transform(frames[b])
        msgs = []
        if len(exp) != len(cands[b]):
            return [(b, "npk", len(exp), len(cands[b]))]
        for j, (a, e) in enumerate(zip(cands[b], exp)):
            for k in ("m_type", "shift"):
                if int(a[k]) != int(e[k]):
                    msgs.append((b, j, k, int(a[k]), int(e[k])))
            if np.float32(a["freq"]).tobytes() != np.float32(e["freq"]).tobytes():
                msgs.append((b, j, "freq", float(a["freq"]), float(e["freq"])))
            if np.float32(a["sync"]).tobytes() != np.float32(e["sync"]).tobytes():
                msgs.append((b, j, "sync", float(a["sync"]), float(e["sync"])))
            if np.float32(a["snr"]).tobytes() != np.float32(e["snr"]).tobytes():      # (round 5: the kernel restates this libm's log10f)
                msgs.append((b, j, "snr", float(a["snr"]), float(e["snr"])))
            if int(e["m_type"]) == 1 and any(a[k] != e[k] for k in ("V1", "V2", "p1", "p2")):
                msgs.append((b, j, "slm"))
            if j < per:
                d = O.demod_candidate(e, 1500, frames[b])
                for k in O.record_diff(out[b, j], d):
                    msgs.append((b, j, k))
        return msgs

    t0 = time.time()
    with ThreadPoolExecutor(nw) as ex:
        for m in ex.map(check, [(i % nw, i) for i in range(n)]):
            bad += m
    tc = time.time() - t0
    ncand = sum(len(c) for c in cands)
    print("frames %d, candidates %d (max %d per frame), GPU %.2f s (host pointers), oracle %.1f s on %d threads"
          % (n, ncand, max(len(c) for c in cands), tg, tc, nw))
    print("mismatches:", len(bad))
    for m in bad[:20]:
        print("  ", m)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
