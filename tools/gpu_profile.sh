#!/bin/bash
# Round measurement set (run on the GPU box): bench line, rocprofv3 kernel stats of the same command, and the
# HBM-traffic counters in separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots).
set -o pipefail
tag=${1:-r1}
envid=${2:-GoalContinuous3P-v0}
batch=${3:-65536}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --env $envid --batch $batch --steps 1000 --warmup 100 2>$out/bench.err | tee $out/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --env $envid --batch $batch --steps 1000 --warmup 100 --no-cpu-baseline --no-kernel-timing"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
ls $out
