mkdir -p gpurun_out/r4
for m in 0 1 2 4 8 16 31 5 0; do
  UWSPR_OPTIONS=stage_kernels=$([ $m = 0 ] && echo 1 || echo 2),rows_mask=$m timeout -k 10 300 python3 bench.py --no-cpu --no-sweep --no-host-legs --no-lazy --repeats 3 > gpurun_out/r4/ab_rows_$m.json 2>/dev/null
  echo -n "rows mask $m: "; python3 tools/bench_brief.py gpurun_out/r4/ab_rows_$m.json | head -1
done
