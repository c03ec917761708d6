#!/usr/bin/env python3
"""Diagnostic: per-wave timeline of the last k4_group launch of one pipeline batch.
Build with UWSPR_EXTRA_HIPFLAGS=-DK4_STAMPS (tools/k4_stamps.sh does)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import gr_uwspr_amd as G

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = G.native
dev = torch.device("cuda", 0)
frames = G.synth.make_frames_torch(B, dev, seed=1, snr_db=-20.0)
ctx = G.Context()
cands = torch.empty(B * 200 * 48, dtype=torch.uint8, device=dev)
npk = torch.empty(B, dtype=torch.int32, device=dev)
out = torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
for _ in range(3):
    ctx.pipeline_batch_into(frames, cands, npk, out, max_per_frame=1)
ctx.synchronize()
L = N.lib()
NLS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for label, nw in (("last k4_group<%d> launch" % NLS, B * (3 if NLS == 6 else 1) * 162 // 16),):
    nw = min(nw, 16384)
    buf = np.zeros((nw, 4), np.uint64)
    rc = L.uwspr_debug_k4_stamps(C.c_void_p(buf.ctypes.data), nw)
    assert rc == 0
    t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64); t2 = buf[:, 2].astype(np.int64)
    ok = t2 > 0
    t0, t1, t2, hw = t0[ok], t1[ok], t2[ok], buf[ok, 3]
    base = t0.min()
    tick = 0.01  # us per wall_clock64 tick (100 MHz)
    print(label, "waves", ok.sum())
    print("  first start 0, last start %.1f us, first end %.1f us, last end %.1f us" %
          ((t0.max() - base) * tick, (t2.min() - base) * tick, (t2.max() - base) * tick))
    life = (t2 - t0) * tick
    pro = (t1 - t0) * tick
    print("  lifetime us: min %.1f median %.1f p90 %.1f max %.1f; prologue median %.1f max %.1f" %
          (life.min(), np.median(life), np.percentile(life, 90), life.max(), np.median(pro), pro.max()))
    hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 15
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    u, cnt = np.unique(cuid, return_counts=True)
    print("  CUs used %d; waves per CU: min %d median %d max %d" % (u.size, cnt.min(), np.median(cnt), cnt.max()))
    su, scnt = np.unique(cuid * 4 + simd, return_counts=True)
    print("  SIMDs used %d; waves per SIMD histogram:" % su.size, np.bincount(scnt))
    # start-time histogram in 5 us bins
    h = np.bincount(((t0 - base) * tick / 5).astype(int))
    print("  wave starts per 5 us bin:", h.tolist())
    h = np.bincount(((t2 - base) * tick / 5).astype(int))
    print("  wave ends   per 5 us bin:", h.tolist())
