#!/bin/bash
# Timing experiments on the fused schedule kernel: -DK6_EXP=<bits> builds (1: no global loads,
# 2: no staging stores / barriers, 4: no scalar phasor loads, 8: no arithmetic) go to
# lib/libuwspr_hip_exp.so (never the product library); results are INVALID, only the phase times count.
for e in "$@"; do
  export UWSPR_EXTRA_HIPFLAGS="-DK6_EXP=$e"
  echo "== K6_EXP=$e"
  python tools/sched_stamps.py 256 2>&1 | tail -11
done
