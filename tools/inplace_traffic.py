#!/usr/bin/env python3
"""Frames cut out of the stream (uwspr_stream_take: 256 x 45000 samples = 92 MB, contiguous) against the
same frames read IN PLACE (uwspr_stream_take_view + uwspr_set_frame_stride(3375): 7.2 MB of unique
samples) -- the workload of one measurement pass.  Run under rocprofv3:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/inplace_traffic.py cut|view [steps]

and sum FETCH_SIZE per kernel from the counter CSV (profiles/r03_inplace_traffic.txt is such a summary).  The stream: noise at -20 dB with a transmission
starting every 45000 samples (every window sees one; one window in 13 sees it inside the search range).
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_uwspr_amd as G

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, hop, fl = 256, 3375, 45000
rng = np.random.default_rng(9)
sig = G.synth.make_frames(20, seed=99, snr_db=None)[:, 375:375 + 162 * 256]
ctx = G.Context()
ctx.stream_open(hop, B)
pinned = torch.empty((B * hop, 2), dtype=torch.float32).pin_memory()
cands = torch.empty(B * ctx.maxfreqs * 48, dtype=torch.uint8, device="cuda")
npk = torch.empty(B, dtype=torch.int32, device="cuda")
out = torch.empty(B * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
cut = torch.empty((B, fl, 2), dtype=torch.float32, device="cuda")


def chunk(k):
    buf = (G.synth.sigma_for_snr(-20.0) * rng.standard_normal((B * hop, 2))).astype(np.float32)
    for t in range(B * hop // 45000 - 1):
        s0 = t * 45000 + int(rng.integers(0, 3000))
        buf[s0:s0 + sig.shape[1]] += sig[(t + k) % 20]
    return buf


ctx.stream_push(chunk(0)[:fl - hop])
for k in range(steps):
    pinned.copy_(torch.from_numpy(chunk(k + 1)))
    ctx.stream_push(pinned.numpy())
    if mode == "cut":
        ctx.stream_take(B, cut)
        ctx.pipeline_batch_into(cut, cands, npk, out, max_per_frame=1)
    else:
        ptr, stride, _ = ctx.stream_take_view(B)
        ctx.set_frame_stride(stride)
        ctx.pipeline_batch_into(G.FrameView(B, ptr=ptr), cands, npk, out, max_per_frame=1)
        ctx.set_frame_stride(0)
    ctx.synchronize()
print(mode, "worth", int(np.frombuffer(out.cpu().numpy().tobytes(), G.native.DEMOD_DTYPE)["worth_a_try"].sum()), "of", B)
ctx.close()
