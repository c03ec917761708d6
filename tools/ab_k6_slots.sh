# Round 6, review item 3 (GPU box): the fused kernel as 8-wavefront workgroups, two per CU (experiment build -DK6_SLOTS=2,
# option sched_grid = 512), against the library's 16-wavefront form, alternating, 1 / 2 / 3 streams; first its bytes against
# the staged form and the oracle (the equivalence and schedule tests run inside the experiment library).
#   bash tools/ab_k6_slots.sh  -> gpurun_out/ab_k6_slots/summary.txt
O=gpurun_out/ab_k6_slots
rm -rf $O; mkdir -p $O
EXP="-DK6_SLOTS=2"
UWSPR_EXTRA_HIPFLAGS="$EXP" python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
UWSPR_EXTRA_HIPFLAGS="$EXP" UWSPR_OPTIONS=sched_grid=512 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q \
    -k "forms or schedule or pipeline or lazy" > $O/parity_exp.log 2>&1
echo "parity (experiment library, sched_grid=512): $(tail -1 $O/parity_exp.log)" | tee $O/summary.txt
UWSPR_EXTRA_HIPFLAGS="$EXP" UWSPR_OPTIONS=sched_grid=512 timeout -k 10 300 python3 tools/soak_parity.py 600 10 0 139000 > $O/soak_exp.log 2>&1
grep -E "^options|^frames|mismatches" $O/soak_exp.log | tee -a $O/summary.txt
for rep in 1 2; do
  for ns in 1 2 3; do
    for form in lib exp; do
      if [ $form = exp ]; then export UWSPR_EXTRA_HIPFLAGS="$EXP"; export UWSPR_OPTIONS=sched_grid=512; else unset UWSPR_EXTRA_HIPFLAGS; unset UWSPR_OPTIONS; fi
      timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu --no-sweep --no-lazy --no-host-legs --sched fused --streams $ns \
          > $O/${form}_x${ns}_r$rep.json 2> $O/${form}_x${ns}_r$rep.err
      python3 - $form $ns $O/${form}_x${ns}_r$rep.json <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("%-4s fused x %s streams: value %8.0f  ms/step %.4f  k6 alone %.1f us  spot check %s" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], 1e3 * d["roofline"]["kernel_ms_per_step"],
    (d.get("parity_spot_check") or {}).get("equal")))
PY
    done
  done
done
unset UWSPR_EXTRA_HIPFLAGS; unset UWSPR_OPTIONS
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu --no-sweep --no-lazy --no-host-legs --sched staged --streams 3 > $O/staged_x3.json 2> $O/staged_x3.err
python3 -c "
import json; d = json.loads(open('$O/staged_x3.json').read().strip().splitlines()[-1]); print('lib  staged x 3 streams: value %8.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/summary.txt
