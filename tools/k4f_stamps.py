#!/usr/bin/env python3
"""Diagnostic (GPU box): where the cycles of a k4_fpack wavefront go -- barrier waits, chunk store + phasor
generation, multiply-add walk -- and how many wavefronts share a SIMD, for the LAST k4_fpack launch (S4) of a
256-frame batch.  Build: UWSPR_EXTRA_HIPFLAGS=-DK4F_STAMPS (own library file)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import os
os.environ.setdefault("UWSPR_OPTIONS", "sched=0")
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
nw = min(4096, (B * 162 + 63) // 64 * 4)
buf = np.zeros((nw, 10), np.uint64)
assert L.uwspr_debug_k4f_stamps(C.c_void_p(buf.ctypes.data), nw) == 0
b = buf.astype(np.int64)
ok = b[:, 1] > 0
b = b[ok]
life = b[:, 1] - b[:, 0]
real = (b[:, 3] - b[:, 2]) / 100.0   # us
clk = life / np.maximum(real, 1e-9)  # MHz
span = (b[:, 3].max() - b[:, 2].min()) / 100.0
print("waves %d  tabled %d  kernel span %.1f us  wave lifetime us: min %.1f med %.1f max %.1f  clock med %.0f MHz" %
      (len(b), int(b[:, 9].sum()), span, real.min(), np.median(real), real.max(), np.median(clk)))
for name, col in (("barrier 1 wait", 4), ("store + generator", 5), ("barrier 2 wait", 6), ("multiply-add walk", 7)):
    v = b[:, col]
    print("  %-18s cycles/wave: med %7d  mean %7d  (%.1f %% of lifetime)" % (name, np.median(v), v.mean(), 100.0 * v.sum() / life.sum()))
print("  other (prologue/epilogue)     %.1f %%" % (100.0 * (life.sum() - b[:, 4:8].sum()) / life.sum()))
hw = b[:, 8]
hwid = hw & 0xFFFFFFFF
xcc = (hw >> 32) & 0xF
simd = (hwid >> 4) & 3
cuid = ((xcc * 8 + ((hwid >> 13) & 7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 15)
su, scnt = np.unique(cuid * 4 + simd, return_counts=True)
print("  SIMDs used %d; wavefronts per SIMD histogram:" % su.size, np.bincount(scnt).tolist())
# per-SIMD: walk cycles per instruction by the number of waves on the SIMD
key = cuid * 4 + simd
cnt = dict(zip(su.tolist(), scnt.tolist()))
per = np.array([cnt[k] for k in key.tolist()])
for n in sorted(set(per.tolist())):
    m = per == n
    print("  waves on a SIMD with %d: lifetime med %.1f us, walk med %d cycles, end time med %.1f us" %
          (n, np.median(real[m]), np.median(b[m, 7]), np.median((b[m, 3] - b[:, 2].min()) / 100.0)))
tone = np.arange(len(ok))[ok] % 4
for t in range(4):
    m = tone == t
    print("  tone wave %d: bar1 %d gen %d bar2 %d walk %d" % (t, np.median(b[m, 4]), np.median(b[m, 5]), np.median(b[m, 6]), np.median(b[m, 7])))
t_start = (b[:, 2] - b[:, 2].min()) / 100.0
t_end = (b[:, 3] - b[:, 2].min()) / 100.0
print("  wave starts per 4 us bin:", np.bincount((t_start / 4).astype(int)).tolist())
print("  wave ends   per 4 us bin:", np.bincount((t_end / 4).astype(int)).tolist())
late = t_end > np.percentile(t_end, 90)
print("  latest 10 %% of the waves: start med %.1f us, lifetime med %.1f us, walk med %d, gen med %d, bar2 med %d; on SIMDs with %s waves" %
      (np.median(t_start[late]), np.median(real[late]), np.median(b[late, 7]), np.median(b[late, 5]), np.median(b[late, 6]),
       np.bincount(per[late]).tolist()))
wg_per_cu = np.bincount(np.unique(cuid, return_counts=True)[1] // 4)
print("  workgroups per CU histogram:", wg_per_cu.tolist())
