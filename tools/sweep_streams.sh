# bench rate vs number of streams and HW queues (GPU box)
set -e
O=gpurun_out/streams
mkdir -p $O
for q in 4 8; do
for n in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep --streams $n > $O/q${q}_s$n.log 2>&1
done
done
