# per-kernel average durations of the staged form, single stream (GPU box):
#   tools/prof_staged_kernels.sh NAME   (environment switches are inherited) -> gpurun_out/pk_NAME.txt
set -e
name=$1
D=gpurun_out/pk_$name
rm -rf $D; mkdir -p $D
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 40 --warmup 5 --no-cpu --no-sweep --no-lazy \
    --no-host-legs --sched staged --streams 1 --repeats 1 > $D/bench.json 2> $D/bench.err
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/pk_$name.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("%-60s calls %6s  avg %8.2f us  total %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["TotalDurationNs"]) / 1e6))
PY
cat gpurun_out/pk_$name.txt
