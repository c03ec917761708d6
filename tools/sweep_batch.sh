# bench rate vs frames per step and streams (GPU box)
set -e
O=gpurun_out/batch
mkdir -p $O
for cfg in "256 3" "512 1" "512 2" "768 1" "768 2" "1024 1" "1024 2" "2048 1" "2048 2"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 120 --warmup 12 --no-cpu --no-sweep --frames $1 --streams $2 > $O/f$1_s$2.log 2>&1
done
