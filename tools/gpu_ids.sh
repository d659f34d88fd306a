#!/bin/bash
# measurement aid: rollout throughput of every BASELINE config / id family on one GPU, and the batch scaling of 3P
run() { timeout -k 10 200 python bench.py --env $1 --batch $2 --steps $3 --warmup 50 --chunk $3 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print('%-24s B=%-8d %-42s us/step %7.3f  G env-steps/s %6.2f  frac %.3f  per-step-launch us/step %.2f' % ('$1', $2, r['kernel'], b['ms_per_step']*1e3, b['value']/1e9, r['frac'], b['ms_per_step_one_launch_per_step']*1e3))"; }
run GoalContinuous2P-v0 4096 1000
run GoalContinuous2P-v0 65536 1000
run GoalContinuous3P-v0 65536 1000
run GoalContinuous4P-v0 65536 1000
run KeplerCircleOrbit-v0 65536 1000
run KeplerEllipseHard-v0 65536 1000
run KeplerRandomOrbits-v0 65536 1000
run GoalContinuous3P-v0 131072 600
run GoalContinuous3P-v0 196608 400
run GoalContinuous3P-v0 262144 400
run GoalContinuous3P-v0 524288 300
run GoalContinuous3P-v0 1048576 200
