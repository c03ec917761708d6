#!/bin/bash
# Runs on the GPU box (gpurun): kernel traces + separate PMC passes of bench.py for BOTH schedule forms
# (fused k6_sched / staged k4_* + k5_*), single stream, plus the default bench line.  Raw CSVs land
# under gpurun_out/prof_$1; tools/make_final_profile.sh condenses them into profiles/.
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import gr_uwspr_amd as G; print(G.native.source_digest())" > $OUT/library_sources_sha256.txt   # what the counters belong to
hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null
for form in fused staged; do
  B="python3 bench.py --no-cpu --no-sweep --no-lazy --no-host-legs --sched $form --streams 1 --repeats 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$form -- $B --steps 20 --warmup 3 > $OUT/trace_$form.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$form -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch_$form.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$form -- $B --steps 3 --warmup 1 > $OUT/pmc_write_$form.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$form -- $B --steps 3 --warmup 1 > $OUT/pmc_sq_$form.log 2>&1
  echo "$form done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $OUT/trace_default.log 2>&1
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- /tmp/fetch_calib > $OUT/calib.log 2>&1
timeout -k 10 400 python3 bench.py > $OUT/bench_plain.log 2>&1
echo done
