# bench rate vs number of streams (GPU box)
set -e
O=gpurun_out/streams
mkdir -p $O
for n in 2 3 4 5; do
  timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep --streams $n > $O/s$n.log 2>&1
done
