# A/B of two builds on the default 3-stream bench, alternating (GPU box): ab_flags2.sh "<flagsA>" "<flagsB>"
set -e
O=gpurun_out/ab_flags2
mkdir -p $O
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export UWSPR_EXTRA_HIPFLAGS="$1"; else export UWSPR_EXTRA_HIPFLAGS="$2"; fi
  python3 -c "import gr_uwspr_amd as G; G.build()" 2>/dev/null
  timeout -k 10 200 python3 bench.py --steps 240 --warmup 24 --no-cpu --no-sweep > $O/${v}_$rep.log 2>&1
done
done
