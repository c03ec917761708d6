import sys
sys.path.insert(0, '/root/repo')
import torch
import gr_uwspr_amd as G
fr = G.synth.make_frames_torch(256, "cuda", snr_db=-20.0)
ctx = G.Context()
for _ in range(3):
    ctx.pipeline_batch(fr, max_per_frame=1, fetch=False)
ctx.synchronize()
