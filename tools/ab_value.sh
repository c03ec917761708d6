# A/B of environment settings on the timed configuration (GPU box):
#   tools/ab_value.sh NAME "ENV1=a ENV2=b" "ENV1=c" ...   -> gpurun_out/ab_NAME/*.json, one summary line per setting
# every setting runs twice, interleaved, staged x3 streams (what `value` is measured on) unless SCHED/STREAMS say otherwise
set -e
name=$1; shift
O=gpurun_out/ab_$name
mkdir -p $O
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    env $setting timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu --no-sweep --no-lazy --no-host-legs \
        --sched ${SCHED:-staged} --streams ${STREAMS:-3} > $O/s${i}_r$rep.json 2> $O/s${i}_r$rep.err
    python3 - "$setting" $O/s${i}_r$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
print("%-40s value %8.0f  ms/step %.4f  1-stream kernels: %s" % (sys.argv[1], d["value"], d["ms_per_step"],
      " ".join("%s %.1f" % (n[:5], 1e3 * v["ms_per_step"]) for n, v in k.items())))
PY
  done
done
