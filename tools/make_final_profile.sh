#!/bin/bash
# Condenses gpurun_out/prof_$1 (written by tools/run_profiles.sh) into the committed
# profiles/$2_* files.  Usage: tools/make_final_profile.sh r01i r01_final
set -e
P=gpurun_out/prof_$1
O=profiles/$2
H=$(git rev-parse --short HEAD)
{
echo "# build at commit $H (+ working tree); MI355X gfx950, ROCm 7.2; script: tools/run_profiles.sh"
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu   (default 3 streams; pipeline steps + configs[2] sweep: grid and flat forms)"
python3 tools/prof_summary.py $P/trace
echo
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --streams 1   (launch durations without overlap: compare roofline.single_stream)"
python3 tools/prof_summary.py $P/trace_streams1
echo
echo "# separate PMC passes, bench.py --steps 3 --warmup 1 --no-cpu; FETCH_SIZE/WRITE_SIZE in KB per launch, UNCORRECTED"
python3 tools/prof_summary.py $P/pmc_fetch | sed -n '/counters/,$p'
python3 tools/prof_summary.py $P/pmc_write | sed -n '/counters/,$p'
python3 tools/prof_summary.py $P/pmc_sq | sed -n '/counters/,$p'
echo
echo "# FETCH_SIZE calibration (tools/fetch_calib.hip)"
cat $P/calib.log
python3 tools/prof_summary.py $P/calib_fetch | sed -n '/counters/,$p'
} > ${O}_rocprof_summary.txt
grep "^{" $P/bench_plain.log > ${O}_bench.json
cp $(ls $P/trace/*/*kernel_stats.csv | head -1) ${O}_kernel_stats.csv
cp $(ls $P/trace_streams1/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_streams1.csv
wc -l ${O}_rocprof_summary.txt
