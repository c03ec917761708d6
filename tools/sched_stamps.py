"""Phase times inside the fused schedule kernel (k6_sched): option sched_stamps = 1, build-free
diagnostic.  Prints the median / max duration of every phase over the candidates of one batch."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["UWSPR_OPTIONS"] = "sched=1,sched_stamps=1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = G.synth.make_frames_torch(B, "cuda", snr_db=-20.0)
ctx = G.Context()
for _ in range(3):
    ctx.pipeline_batch(frames, max_per_frame=1, fetch=False)
ctx.synchronize()
L = G.native.lib()
st = np.zeros((B, 64), np.uint64)
L.uwspr_debug_sched_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = L.uwspr_debug_sched_stamps(ctx.h, st.ctypes.data, B)
assert rc == 0, rc
t = st.astype(np.int64)
names = ["tables A", "S0", "S1", "S2", "gate", "tables B", "S3", "S4", "S5 pass", "S5 fold"]
d = (t[:, 1:10] - t[:, 0:9]) / 100.0   # 100 MHz ticks -> us
print("phase          median_us   max_us")
names = ["tables A", "S0", "S1", "S2", "tables B", "S3", "S4", "S5 pass", "S5 fold"]
for k, n in enumerate(names):
    print("%-12s %9.2f %9.2f" % (n, np.median(d[:, k]), d[:, k].max()))
print("S1 pass     %9.2f   S1 fold %9.2f   S1 rest %9.2f" % (np.median(t[:, 10] - t[:, 2]) / 100.0,
      np.median(t[:, 11] - t[:, 10]) / 100.0, np.median(t[:, 3] - t[:, 11]) / 100.0))
tot = (t[:, 9] - t[:, 0]) / 100.0
print("total        %9.2f %9.2f" % (np.median(tot), tot.max()))
print("launch span  %9.2f us (first start to last end)" % ((t[:, 9].max() - t[:, 0].min()) / 100.0))
