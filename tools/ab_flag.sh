# A/B of a compile flag on the default bench + S-stage kernel times (GPU box): tools/ab_flag.sh "<flags>" tag
set -e
export TMPDIR=/tmp
O=gpurun_out/ab_flag_$2
mkdir -p $O
export UWSPR_EXTRA_HIPFLAGS="$1"
python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu --no-sweep --streams 1 > $O/trace.log 2>&1
timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep > $O/bench.log 2>&1
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu > $O/pytest.log 2>&1
