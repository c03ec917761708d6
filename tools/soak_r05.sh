# round-5 soak on the FINAL build (k4_jig for S5 incl. its per-lane recurrence walk, restated log10f in K2): whole pipeline,
# every candidate, GPU vs the CPU oracle record by record (tools/soak_parity.py), seeds no earlier run used (base 121000)
set -x
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 3500 10 0 121000                      # staged, defaults: weak / noise frames whose candidates stage 2 gives a drift
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1500 40 2 122000                      # ... drifting candidates from the FDR
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1000 20 4 123000
UWSPR_OPTIONS=sched=0,reuse=0 python3 tools/soak_parity.py 1000 10 0 124000              # without the stage-winner reuse (all 17 tries through k4_jig)
UWSPR_OPTIONS=sched=0,k4_forms=0 python3 tools/soak_parity.py 1000 10 0 125000           # S5 through the LDS-ring kernel
python3 tools/soak_parity.py 1500 10 0 126000                                            # fused (default)
