for cfg in "256 3" "256 4" "512 2" "512 3" "384 3"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu --no-sweep --no-lazy --no-host-legs --sched staged --streams $2 --frames $1 > gpurun_out/sw_$1_$2.json 2> gpurun_out/sw_$1_$2.err
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/sw_$1_$2.json')); print('frames $1 streams $2 value %.0f ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
