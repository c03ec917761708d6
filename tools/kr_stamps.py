#!/usr/bin/env python3
"""Diagnostic (GPU box): where the cycles of a k4_rows wavefront go (prologue, walk, chunk store + barrier) and how
the wavefronts sit on the SIMDs, for the launch of the stage compiled in with -DKR_STAMP_KIND=<0..4> (default S1) of a
256-frame batch.  Build: UWSPR_EXTRA_HIPFLAGS=-DKR_STAMPS[,-DKR_STAMP_KIND=n] (own library file); staged schedule."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
os.environ["UWSPR_OPTIONS"] = os.environ.get("UWSPR_OPTIONS", "sched=0,stage_kernels=2")
import gr_uwspr_amd as G

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = G.native
dev = torch.device("cuda", 0)
frames = G.synth.make_frames_torch(B, dev, seed=1, snr_db=-20.0)
ctx = G.Context()
cands = torch.empty(B * 200 * 48, dtype=torch.uint8, device=dev)
npk = torch.empty(B, dtype=torch.int32, device=dev)
out = torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
for _ in range(5):
    ctx.pipeline_batch_into(frames, cands, npk, out, max_per_frame=1)
ctx.synchronize()
L = N.lib()
nw = 8192
buf = np.zeros((nw, 8), np.uint64)
assert L.uwspr_debug_kr_stamps(C.c_void_p(buf.ctypes.data), nw) == 0
b = buf.astype(np.int64)
b = b[b[:, 1] > 0]
life = b[:, 1] - b[:, 0]
real = (b[:, 3] - b[:, 2]) / 100.0
span = (b[:, 3].max() - b[:, 2].min()) / 100.0
print("waves %d  kernel span %.1f us  wave lifetime us: min %.1f med %.1f max %.1f  clock med %.0f MHz" %
      (len(b), span, real.min(), np.median(real), real.max(), np.median(life / np.maximum(real, 1e-9))))
for name, col in (("prologue", 4), ("walk", 5), ("store + barrier", 6)):
    v = b[:, col]
    print("  %-16s cycles/wave: med %7d  mean %7d  (%.1f %% of lifetime)" % (name, np.median(v), v.mean(), 100.0 * v.sum() / life.sum()))
print("  epilogue          %.1f %%" % (100.0 * (life.sum() - b[:, 4:7].sum()) / life.sum()))
hw = b[:, 7]
hwid = hw & 0xFFFFFFFF
xcc = (hw >> 32) & 0xF
cuid = ((xcc * 8 + ((hwid >> 13) & 7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 15)
su, scnt = np.unique(cuid * 4 + ((hwid >> 4) & 3), return_counts=True)
print("  SIMDs used %d; wavefronts per SIMD histogram:" % su.size, np.bincount(scnt).tolist())
cu, ccnt = np.unique(cuid, return_counts=True)
print("  CUs used %d; wavefronts per CU histogram:" % cu.size, np.bincount(ccnt).tolist())
t_start = (b[:, 2] - b[:, 2].min()) / 100.0
t_end = (b[:, 3] - b[:, 2].min()) / 100.0
print("  wave starts per 4 us bin:", np.bincount((t_start / 4).astype(int)).tolist())
print("  wave ends   per 4 us bin:", np.bincount((t_end / 4).astype(int)).tolist())
# ---- per-chunk timeline of the wavefronts of ONE CU
tl = np.zeros((nw, 16, 4), np.uint64)
if hasattr(L, "uwspr_debug_kr_timeline") and L.uwspr_debug_kr_timeline(C.c_void_p(tl.ctypes.data), nw) == 0:
    tl = tl.astype(np.int64)
    full = buf.astype(np.int64)
    okw = np.nonzero(full[:, 1] > 0)[0]
    hw_all = full[okw, 7]
    hwid_all = hw_all & 0xFFFFFFFF
    cu_all = (((hw_all >> 32) & 0xF) * 8 + ((hwid_all >> 13) & 7)) * 32 + ((hwid_all >> 12) & 1) * 16 + ((hwid_all >> 8) & 15)
    simd_all = (hwid_all >> 4) & 3
    pick = cu_all[0]
    m = cu_all == pick
    sel = okw[m]; simd = simd_all[m]
    t00 = full[sel, 0].min()
    clk = float(np.median(life / np.maximum(real, 1e-9)))   # MHz
    print("timeline of one CU (%d wavefronts), us from the first wave's start" % len(sel))
    print("per workgroup and chunk: arithmetic [first wave out of the turn .. last wave into the next turn], then the turn: "
          "wait+store / barrier wait / load issue (medians over the workgroup's waves, us)")
    for g, (wgw, sd) in enumerate(zip(np.array_split(sel, 3), np.array_split(simd, 3))):
        line = []
        for c in range(16):
            if tl[wgw, c, 3].max() == 0:
                continue
            nxt = tl[wgw, c + 1, 0] if c + 1 < 16 and tl[wgw, c + 1, 3].max() > 0 else full[wgw, 1]
            ar = (nxt - tl[wgw, c, 3]) / clk
            line.append("c%d %.1f-%.1f arith/wave %.2f..%.2f | %.2f/%.2f/%.2f" % (
                c, (tl[wgw, c, 3].min() - t00) / clk, (nxt.max() - t00) / clk, ar.min(), ar.max(),
                np.median(tl[wgw, c, 1] - tl[wgw, c, 0]) / clk, np.median(tl[wgw, c, 2] - tl[wgw, c, 1]) / clk,
                np.median(tl[wgw, c, 3] - tl[wgw, c, 2]) / clk))
        print("  WG %d (SIMDs %s):" % (g, "".join(str(int(x)) for x in sd)))
        for x in line:
            print("     " + x)
