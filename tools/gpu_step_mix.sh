#!/bin/bash
# measurement aid (run on the GPU box): dynamic instruction counts and issue-slot occupancy of the one-launch-per-step kernel, separate --pmc passes
#   tools/gpu_step_mix.sh TAG ENV_ID BATCH [single|pair]
tag=${1:-stepmix}; envid=${2:-GoalContinuous3P-v0}; batch=${3:-1048576}; plan=${4:-single}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/gpu_step_times.py $envid $batch $plan"
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
  "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_ICACHE_REQ SQC_DCACHE_REQ SQC_TC_REQ" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- $B > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "step_kernel" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/mix.txt", "w") as fo:
    for k, d in acc.items():
        fo.write("kernel %s\n" % k)
        for c in sorted(d):
            v = sorted(d[c])
            fo.write("  %-28s median-dispatch %.5g (n=%d)\n" % (c, v[len(v) // 2], len(v)))
print(open(out + "/mix.txt").read())
PY
