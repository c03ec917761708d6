#!/bin/bash
# Condenses gpurun_out/prof_$1 (written by tools/run_profiles.sh) into the committed
# profiles/$2_* files and profiles/k4_traffic.json.  Usage: tools/make_final_profile.sh r02 r02_final
set -e
P=gpurun_out/prof_$1
O=profiles/$2
H=$(git rev-parse --short HEAD)
{
echo "# build at commit $H (+ working tree); MI355X gfx950, ROCm 7.2; script: tools/run_profiles.sh"
for form in fused staged; do
echo
echo "# ===== schedule form: $form ====="
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-sweep --no-lazy --no-host-legs --sched $form --streams 1 --repeats 1 --steps 20 --warmup 3"
python3 tools/prof_summary.py $P/trace_$form
echo "# separate PMC passes (--steps 3 --warmup 1); FETCH_SIZE/WRITE_SIZE in KB per launch, UNCORRECTED"
python3 tools/prof_summary.py $P/pmc_fetch_$form | sed -n '/counters/,$p'
python3 tools/prof_summary.py $P/pmc_write_$form | sed -n '/counters/,$p'
python3 tools/prof_summary.py $P/pmc_sq_$form | sed -n '/counters/,$p'
done
echo
echo "# ===== default bench.py (form and streams by its own trial) ====="
python3 tools/prof_summary.py $P/trace_default
echo
echo "# FETCH_SIZE calibration (tools/fetch_calib.hip)"
cat $P/calib.log
python3 tools/prof_summary.py $P/calib_fetch | sed -n '/counters/,$p'
} > ${O}_rocprof_summary.txt
grep "^{" $P/bench_plain.log > ${O}_bench.json
for form in fused staged; do cp $(ls $P/trace_$form/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_${form}_streams1.csv; done
cp $(ls $P/trace_default/*/*kernel_stats.csv | head -1) ${O}_kernel_stats.csv
python3 tools/make_traffic.py $P ${O}_rocprof_summary.txt
# the bench line was printed before this run's PMC passes were condensed: give it this run's figures
python3 - ${O}_bench.json <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
t = json.load(open("profiles/k4_traffic.json"))
form = "fused" if "fused" in b["config"].get("sched", "") else "staged"
b["roofline"]["traffic"] = t[form]["bytes_per_step"]
b["roofline"]["valu_issue_utilisation_pmc"] = t[form]["valu_issue_utilisation"]
b["roofline"]["traffic_source"] = t[form]["source"]
b["roofline"]["traffic_stale"] = None   # the counters are this run's own
open(sys.argv[1], "w").write(json.dumps(b) + "\n")
PY
wc -l ${O}_rocprof_summary.txt
