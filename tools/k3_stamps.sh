# diagnostic: K3 phase timings (GPU box)
set -e
export UWSPR_EXTRA_HIPFLAGS="-DK3_STAMPS"
python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
python3 - <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import gr_uwspr_amd as G
N = G.native
dev = torch.device("cuda", 0)
frames = G.synth.make_frames_torch(256, dev, seed=1, snr_db=-20.0)
ctx = G.Context()
c, n = ctx.fdr_batch(frames) if hasattr(ctx, "fdr_batch") else (None, None)
ctx.synchronize()
PY
