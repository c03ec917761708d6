# diagnostic: per-wave timeline of the last k4_group<5> launch (S0) of a 256-frame batch (GPU box)
set -e
export UWSPR_EXTRA_HIPFLAGS="-DK4_STAMPS -DK4_STAMP_NL=5"
python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
python3 tools/k4_stamps.py 256 5
