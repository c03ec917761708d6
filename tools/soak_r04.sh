# round-4 soak on the final build: whole pipeline, every candidate, GPU vs the CPU oracle (tools/soak_parity.py), on
# seeds no earlier round used (base 91000): every schedule form, three geometries
set -x
python3 tools/soak_parity.py 2000 10 0 91000                                             # fused (default)
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 2000 10 0 92000                       # staged, packed / ring kernels
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1500 40 2 95000                       # packed kernels, drifting candidates (flat-kernel fallback of S0)
UWSPR_OPTIONS=sched=0,stage_kernels=0 python3 tools/soak_parity.py 1000 20 4 96000       # flat kernel everywhere
