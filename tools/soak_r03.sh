# round-3 soak on the final build: whole pipeline, every candidate, GPU vs the CPU oracle (tools/soak_parity.py)
set -x
python3 tools/soak_parity.py 3000 10 0
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 3000 10 0
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1500 40 1
python3 tools/soak_parity.py 1000 20 4
UWSPR_OPTIONS=sched=0,phasor_tables=0 python3 tools/soak_parity.py 1000 20 2
