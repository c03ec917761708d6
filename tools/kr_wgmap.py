import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
os.environ["UWSPR_OPTIONS"] = os.environ.get("UWSPR_OPTIONS", "sched=0,stage_kernels=2")
import gr_uwspr_amd as G
N = G.native; dev = torch.device("cuda", 0)
B = 256
frames = G.synth.make_frames_torch(B, dev, seed=1, snr_db=-20.0)
ctx = G.Context()
cands = torch.empty(B * 200 * 48, dtype=torch.uint8, device=dev); npk = torch.empty(B, dtype=torch.int32, device=dev)
out = torch.empty(B * N.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
for _ in range(3): ctx.pipeline_batch_into(frames, cands, npk, out, max_per_frame=1)
ctx.synchronize()
L = N.lib(); nw = 8192
buf = np.zeros((nw, 8), np.uint64)
assert L.uwspr_debug_kr_stamps(C.c_void_p(buf.ctypes.data), nw) == 0
b = buf.astype(np.int64); b = b[b[:, 1] > 0]
hw = b[:, 7]; hwid = hw & 0xFFFFFFFF
cu = (((hw >> 32) & 0xF) * 8 + ((hwid >> 13) & 7)) * 32 + ((hwid >> 12) & 1) * 16 + ((hwid >> 8) & 15)
for c in np.unique(cu)[:12]:
    m = cu == c
    ids = np.unique(b[m, 4])
    ends = [(int(i), round(float((b[m & (b[:, 4] == i), 3].max() - b[:, 2].min()) / 100.0), 1)) for i in ids]
    print("CU", int(c), "xcc", int(c) >> 8, "blockIdx -> end us:", ends)
cls = b[:, 4] // 256
t_end = (b[:, 3] - b[:, 2].min()) / 100.0
for k in range(3):
    print("blockIdx class", k, "end time med %.1f us" % np.median(t_end[cls == k]))
