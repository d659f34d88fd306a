#!/bin/bash
# measurement aid: steady-state step-kernel time of differently compiled builds of the same library
for v in "" _maxilp _bias0 _maxilp_bias0; do
  lib=$GRAFT_REPO_ROOT/space_gym_amd/lib/libspacegym_hip$v.so
  [ -f $lib ] || continue
  echo "== $lib"
  SPACEGYM_LIB=$lib timeout -k 10 120 python tools/gpu_breakdown.py GoalContinuous3P-v0 65536 2>&1 | grep -E "avg_us|steady|steps" | paste -sd' ' | sed 's/  */ /g'
done
