import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import gr_uwspr_amd as G
fr = G.synth.make_frames(16, seed=3, snr_db=-18.0)
free0 = None
for it in range(120):
    c = G.Context(halfbandwidth=10 + (it % 3) * 10, maxdrift=it % 2)
    cands, out = c.pipeline_batch(fr, max_per_frame=2)
    a = c.frontend(np.zeros((1, 1000), np.float32))
    c.close()
    if it == 10:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print("free after 10 iterations %d MB, after 120 iterations %d MB, delta %d KB" % (free0 >> 20, free1 >> 20, (free0 - free1) >> 10))
assert free0 - free1 < 64 << 20
print("leak check ok")
