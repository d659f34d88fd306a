#!/bin/bash
# measurement aid (run on the GPU box): dynamic instruction mix and issue-slot occupancy of the rollout kernel, separate --pmc passes
tag=${1:-mix}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-kernel-timing ${@:2}"
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VSKIPPED SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
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
        if "rollout" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/mix.txt", "w") as fo:
    for k, d in acc.items():
        fo.write("kernel %s\n" % k)
        for c in sorted(d):
            fo.write("  %-28s max-dispatch %.4g (n=%d)\n" % (c, max(d[c]), len(d[c])))
print(open(out + "/mix.txt").read())
PY
