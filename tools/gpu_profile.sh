#!/bin/bash
# Round measurement set (run on the GPU box): bench line, rocprofv3 kernel stats of the same command, and the
# HBM-traffic counters in separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots).
#   tools/gpu_profile.sh TAG [ENV_ID] [BATCH] [STEPS] [WARMUP]        default: the driver's command (--steps 20 --warmup 5)
set -o pipefail
tag=${1:-r2}
envid=${2:-GoalContinuous3P-v0}
batch=${3:-65536}
steps=${4:-20}
warmup=${5:-5}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --gpus 1 --env $envid --batch $batch --steps $steps --warmup $warmup 2>$out/bench.err | tee $out/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --env $envid --batch $batch --steps $steps --warmup $warmup --no-cpu-baseline --no-host-path"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --no-kernel-timing > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --no-kernel-timing > $out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- $B --no-kernel-timing > $out/pmc_sq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $out $steps > $out/summary.txt 2>&1
cat $out/summary.txt
