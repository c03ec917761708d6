# diagnostic: per-wave timeline of k4_group<5>/<6> for build variants (GPU box)
set -e
for v in "-DK4G_VARIANT=0" ; do
for nl in 5 6; do
export UWSPR_EXTRA_HIPFLAGS="-DK4_STAMPS -DK4_STAMP_NL=$nl $v"
python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
echo "== variant '$v' NL=$nl"
python3 tools/k4_stamps.py 256 $nl
done
done
