#!/bin/bash
# measurement aid: interleaved A/B of differently compiled builds of the same library (same box, 3 rounds)
for round in 1 2 3; do
for v in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/space_gym_amd/lib/libspacegym_hip$v.so
  [ -f $lib ] || continue
  SPACEGYM_LIB=$lib timeout -k 10 200 python bench.py --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s fused us/step %.3f   per-step-launch us/step %.3f' % ('$v' or 'current', b['ms_per_step']*1e3, b['ms_per_step_one_launch_per_step']*1e3))"
done; done
