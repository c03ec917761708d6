# quick A/B helper (GPU box): single-stream kernel trace + default bench for the current build
set -e
export TMPDIR=/tmp
O=gpurun_out/ab_$1
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu --streams 1 > $O/trace.log 2>&1
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu --no-sweep > $O/bench.log 2>&1
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu > $O/pytest.log 2>&1
