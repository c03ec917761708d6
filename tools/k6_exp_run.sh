#!/bin/bash
# runs the prebuilt experiment libraries build/exp_<bits>.so (made by tools/k6_exp.sh's builds here)
python tools/sched_stamps.py 256 2>&1 | tail -13 | grep "S0 \|S1 \|S2 \|S1 pass"
for e in "$@"; do
  cp build/exp_$e.so gr-uwspr_amd/lib/libuwspr_hip_exp.so
  cp build/exp_$e.so.cmd gr-uwspr_amd/lib/libuwspr_hip_exp.so.cmd
  export UWSPR_EXTRA_HIPFLAGS="-DK6_EXP=$e"
  echo "== K6_EXP=$e"
  python tools/sched_stamps.py 256 2>&1 | tail -13 | grep "S0 \|S1 \|S2 \|S1 pass"
done
