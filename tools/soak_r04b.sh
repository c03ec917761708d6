# round-4 soak on the FINAL build (k4_dpair for S2, priority rotation in k4_fpack): whole pipeline, every candidate, GPU vs the
# CPU oracle (tools/soak_parity.py), seeds no earlier run used (base 101000)
set -x
python3 tools/soak_parity.py 1500 10 0 101000                                            # fused (default)
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 3000 10 0 102000                      # staged, packed / ring / pair kernels
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1500 40 2 103000                      # ... drifting candidates: the pair kernel's flat fallback beside it
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1000 20 4 104000
UWSPR_OPTIONS=sched=0,reuse=0 python3 tools/soak_parity.py 1000 10 0 105000              # without the stage-winner reuse (k4_fpack's five-frequency walk)
