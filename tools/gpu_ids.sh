#!/bin/bash
# measurement aid: rollout throughput of every BASELINE config / id family on one GPU, and the batch scaling of 3P
run() { timeout -k 10 200 python bench.py --env $1 --batch $2 --steps $3 --warmup 50 --chunk $3 --no-cpu-baseline --no-host-path ${@:4} 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print('%-24s %-12s B=%-8d %-44s us/step %7.3f  G env-steps/s %6.2f  frac %.3f  per-step-launch us/step %.2f (%s)' % ('$1', b['config'].get('steering', ''), $2, r['kernel'], b['ms_per_step']*1e3, b['value']/1e9, r['frac'], b['ms_per_step_one_launch_per_step']*1e3, b.get('roofline_one_launch_per_step', {}).get('kernel', '')))"; }
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
# discrete-action ids (keyboard_agent.py:10-74) and Steering.acceleration (the classes' constructor default)
run GoalDiscrete3-v0 65536 1000
run KeplerDiscrete-v0 65536 1000
run GoalContinuous3P-v0 65536 1000 --steering acceleration
run KeplerCircleOrbit-v0 65536 1000 --steering acceleration
run GoalContinuous3P-v0 65536 20 --steering acceleration
run GoalDiscrete3-v0 65536 20
