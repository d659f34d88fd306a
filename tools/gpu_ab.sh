#!/bin/bash
# measurement aid (run on the GPU box): interleaved same-box A/B of differently compiled builds of the library --
# space_gym_amd/lib/libspacegym_hip.so ("current") and libspacegym_hip<suffix>.so for every suffix given -- by the kernel time of
# one rollout launch at several steps per launch (tools/gpu_ksweep.py).  Box-to-box differences reach 12 %: only such
# interleaved runs compare builds.
#   tools/gpu_ab.sh ENV_ID BATCH K,K,... [ROUNDS] [suffix ...]      e.g.  tools/gpu_ab.sh GoalContinuous3P-v0 65536 20,1000 2 _old
env=${1:-GoalContinuous3P-v0}; batch=${2:-65536}; ks=${3:-20,1000}; rounds=${4:-2}; shift 4
for round in $(seq 1 $rounds); do
for v in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/space_gym_amd/lib/libspacegym_hip$v.so
  [ -f $lib ] || continue
  SPACEGYM_LIB=$([ -n "$v" ] && echo $lib) timeout -k 10 200 python $GRAFT_REPO_ROOT/tools/gpu_ksweep.py $env $batch $ks 2>&1 | grep "K=" | sed "s/^/$(printf '%-10s' "${v:-current}") /" | cut -c1-120
done; done
