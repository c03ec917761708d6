# A/B of an option on the default bench (GPU box): tools/ab_env.sh NAME   (UWSPR_OPTIONS=NAME=0 against NAME=1)
set -e
O=gpurun_out/ab_env
mkdir -p $O
for rep in 1 2; do
for v in 0 1; do
  UWSPR_OPTIONS=$1=$v timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep > $O/${1}_${v}_$rep.log 2>&1
done
done
