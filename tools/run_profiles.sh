#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + separate PMC passes of bench.py and the
# FETCH_SIZE calibration; raw CSVs land under gpurun_out/prof_$1.
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --streams 1 > $OUT/bench_trace1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_sq.log 2>&1
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- /tmp/fetch_calib > $OUT/calib.log 2>&1
timeout -k 10 300 python3 bench.py > $OUT/bench_plain.log 2>&1
echo done
