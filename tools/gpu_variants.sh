#!/bin/bash
# measurement aid: fused-rollout and per-step timings of differently compiled builds of the same library
for v in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/space_gym_amd/lib/libspacegym_hip$v.so
  [ -f $lib ] || continue
  echo "== $lib"
  SPACEGYM_LIB=$lib timeout -k 10 200 python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused us/step %.3f   per-step-launch us/step %.3f' % (b['ms_per_step']*1e3, b['ms_per_step_one_launch_per_step']*1e3))"
done
