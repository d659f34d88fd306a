#!/bin/bash
# measurement aid: fused-rollout throughput of differently compiled builds at several batch sizes (same box)
for B in 65536 262144 1048576; do
for v in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/space_gym_amd/lib/libspacegym_hip$v.so
  [ -f $lib ] || continue
  SPACEGYM_LIB=$lib timeout -k 10 200 python bench.py --batch $B --steps 200 --warmup 40 --chunk 50 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%-8d %-8s us/step %.3f  G env-steps/s %.2f  frac %.3f' % ($B, '$v' or 'current', b['ms_per_step']*1e3, b['value']/1e9, b['roofline']['frac']))"
done; done
