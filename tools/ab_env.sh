# A/B of an environment switch on the default bench (GPU box): tools/ab_env.sh VAR
set -e
O=gpurun_out/ab_env
mkdir -p $O
for rep in 1 2; do
for v in 0 1; do
  env $1=$v timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep > $O/${1}_${v}_$rep.log 2>&1
done
done
