# per-kernel times (1 stream) + 3-stream bench for a list of flag sets (GPU box): ab_flag_trace.sh "flagsA" "flagsB" ...
set -e
export TMPDIR=/tmp
i=0
for fl in "$@"; do
  O=gpurun_out/abt_$i
  export UWSPR_EXTRA_HIPFLAGS="$fl"
  python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-sweep --streams 1 > $O.log 2>&1
  timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep > ${O}_bench.log 2>&1
  i=$((i+1))
done
