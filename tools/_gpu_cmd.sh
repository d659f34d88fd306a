mkdir -p gpurun_out/r4j
(timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4j/pytest.log; tail -6 gpurun_out/r4j/pytest.log)
for e in GoalContinuous3P-v0 GoalContinuous4P-v0 GoalContinuous2P-v0 KeplerCircleOrbit-v0; do
timeout -k 10 300 python tools/gpu_step_times.py $e 65536,131072,262144,1048576 single,pair 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4j/step_times.txt
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; cut -c1-1500 gpurun_out/r4j/bench.json
