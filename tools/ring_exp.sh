# timing experiments on k4_ring (results invalid, only kernel durations matter) -- GPU box
# K4R_EXP bits: 1 no global loads in the walk, 2 no fences / LDS stores in the walk, 4 no binary64 sincos,
# 8 no prologue loads
set -e
export TMPDIR=/tmp
for e in ${RING_EXPS:-0 1 2 3 4 8}; do
  export UWSPR_EXTRA_HIPFLAGS="-DK4R_EXP=$e"
  python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
  O=gpurun_out/ring_exp_$e
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --streams 1 > $O.log 2>&1
done
