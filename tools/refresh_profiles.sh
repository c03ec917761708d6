#!/bin/bash
# The whole profile refresh on the GPU box after a change under gr-uwspr_amd/csrc or include/ (the library source digest
# changes and bench.py reports `traffic_stale` until profiles/k4_traffic.json is remade):
#   gpurun -- 'bash tools/refresh_profiles.sh r03'     then here:  bash tools/make_final_profile.sh r03 r03_final
# plus profiles/<tag>_valu_lds_by_kernel.txt (from gpurun_out/pmc_all.txt) and <tag>_overlap_3streams.txt (gpurun_out/overlap3.txt).
TAG=${1:-r03}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_all gpurun_out/overlap3   # (also remove them HERE before the call: the merge-back adds to what is there)
bash tools/run_profiles.sh $TAG > gpurun_out/run_profiles.log 2>&1
echo "profiles done"
bash tools/pmc_all.sh > gpurun_out/pmc_all.txt 2>&1
echo "pmc done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/overlap3 -- python3 bench.py --steps 100 --warmup 10 \
    --no-sweep --no-cpu --no-lazy --no-host-legs --sched staged --streams 3 --repeats 3 > gpurun_out/overlap3.log 2>&1
python3 tools/trace_overlap.py gpurun_out/overlap3 > gpurun_out/overlap3.txt 2>&1
find gpurun_out/overlap3 -name "*_kernel_trace.csv" -size +20M -delete   # (keep the merge-back under its size limit)
echo "overlap done"
