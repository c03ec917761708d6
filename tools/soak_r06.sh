# round-6 soak on the FINAL build: whole pipeline, every candidate, GPU vs the CPU oracle record by record
# (tools/soak_parity.py: candidates by PDU field, records incl. all 17 soft-symbol vectors and the jiggered shifts / sync /
# rms), every schedule form, seeds no earlier run used (base 131000).  The oracle's snr is the restated glibc 2.35 log10f.
set -x
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 3000 10 0 131000                      # staged (what bench.py times), defaults
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 1200 40 2 132000                      # ... drifting candidates from the FDR
UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 800 20 4 133000
UWSPR_OPTIONS=sched=0,reuse=0 python3 tools/soak_parity.py 800 10 0 134000               # without the stage-winner reuse
UWSPR_OPTIONS=sched=0,k4_forms=0 python3 tools/soak_parity.py 800 10 0 135000            # S5 through the LDS-ring kernel
UWSPR_OPTIONS=sched=0,k4_forms=3 python3 tools/soak_parity.py 800 10 0 136000            # S0 double-buffered (round 6's experiment)
UWSPR_OPTIONS=sched=0,k4_forms=5 python3 tools/soak_parity.py 800 10 0 137000            # S0 on two wavefronts per tone
python3 tools/soak_parity.py 1500 10 0 138000                                            # fused (default)
# the long leg (argument "long"): 40 000 more frames through the staged form in chunks of 5 000 (1.8 GB of host frames each)
if [ "$1" = "long" ]; then
  for k in 0 1 2 3 4 5 6 7; do
    UWSPR_OPTIONS=sched=0 python3 tools/soak_parity.py 5000 10 0 $((140000 + 1000 * k))
  done
fi
