# A/B/C... of several builds on the default 3-stream bench, alternating (GPU box): ab_flags_n.sh "<flags1>" "<flags2>" ...
# ("-" = the default flags).  One summary line per run in gpurun_out/ab_flags_n/summary.txt.
set -e
O=gpurun_out/ab_flags_n
rm -rf $O; mkdir -p $O
for rep in 1 2; do
  i=0
  for fl in "$@"; do
    i=$((i+1))
    if [ "$fl" = "-" ]; then export UWSPR_EXTRA_HIPFLAGS=""; else export UWSPR_EXTRA_HIPFLAGS="$fl"; fi
    python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
    timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep --no-lazy --no-host-legs > $O/${i}_$rep.log 2>&1
    python3 - "$O/${i}_$rep.log" "$fl" >> $O/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-40s value %7.0f  K4 single stream %.4f ms  frac %.4f  fast %s" % (sys.argv[2], d["value"], r["kernel_ms_per_step"], r["frac"],
      round(d["fast_search"]["frames_per_s"]) if isinstance(d.get("fast_search"), dict) else None))
PY
  done
done
cat $O/summary.txt
