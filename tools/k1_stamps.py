"""Per-wavefront timeline of K1 (spectrogram): needs the stamp build
   UWSPR_EXTRA_HIPFLAGS=-DUWSPR_K1_STAMPS python tools/k1_stamps.py [B]
Prints the wavefront lifetime distribution, the launch span and the number resident over time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batches = [G.synth.make_frames_torch(B, "cuda", snr_db=-20.0, seed=s) for s in range(4)]
ctx = G.Context()
for k in range(8):
    ctx.pipeline_batch(batches[k % 4], max_per_frame=1, fetch=False)
ctx.synchronize()
L = G.native.lib()
nw = min(16384, B * 8 * 4)
st = np.zeros((nw, 3), np.uint64)
L.uwspr_debug_k1_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.uwspr_debug_k1_stamps(st.ctypes.data, nw) == 0
t0 = st[:, 0].astype(np.int64); t1 = st[:, 1].astype(np.int64)
hw = (st[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (st[:, 2] >> np.uint64(32)).astype(np.int64) & 15
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0
life = t1 - t0
work = life > 0.5
print("waves %d  working %d   span %.2f us" % (nw, work.sum(), t1.max()))
print("lifetime us: p5 %.2f p50 %.2f p95 %.2f max %.2f" % tuple(np.percentile(life[work], [5, 50, 95, 100])))
print("start us:    p5 %.2f p50 %.2f p95 %.2f max %.2f" % tuple(np.percentile(t0[work], [5, 50, 95, 100])))
edges = np.arange(0, t1.max() + 2, 2.0)
res = [(int(((t0 <= e) & (t1 > e) & work).sum())) for e in edges]
print("resident working waves every 2 us:", res)
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
print("per-xcc working waves:", np.bincount(xcc[work], minlength=8).tolist())
print("per-xcc last end us:", [round(float(t1[(xcc == x)].max()), 1) if (xcc == x).any() else None for x in range(8)])
key = xcc * 1024 + se * 64 + sh * 16 + cu
u, cnt = np.unique(key[work], return_counts=True)
print("distinct (xcc,se,sh,cu): %d   waves per CU min %d max %d" % (len(u), cnt.min(), cnt.max()))
