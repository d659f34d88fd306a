#!/bin/bash
# measurement aid (run on the GPU box): dynamic instruction counts of the rollout kernel for several builds of the library on
# the same box (same workload: 500-step launches in steady state), separate --pmc passes.  usage: gpu_mix_ab.sh tag lib1 lib2 ...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export SG_KSWEEP_REPS=3
for lib in "$@"; do
  name=$(basename $lib .so)
  export SPACEGYM_LIB=$GRAFT_REPO_ROOT/$lib SPACEGYM_ROLLOUT_KERNEL=pair
  i=0
  for set in \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
    "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_IFETCH SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$name.p$i -- python3 $GRAFT_REPO_ROOT/tools/gpu_ksweep.py GoalContinuous3P-v0 65536 500 > $out/$name.p$i.log 2>&1 || echo "$name pass $i failed"
  done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*.p*/**/*counter_collection.csv", recursive=True):
    lib = os.path.relpath(f, out).split(".p")[0]
    for r in csv.DictReader(open(f)):
        if "pair_rollout" in r["Kernel_Name"]:
            acc[lib][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
with open(out + "/mix_ab.txt", "w") as fo:
    fo.write("per 500-step launch (max over dispatches); per wave-pair and step = / (1024 * 500)\n")
    fo.write("%-28s" % "counter" + "".join("%22s" % n[-20:] for n in names) + "\n")
    for c in sorted({c for n in names for c in acc[n]}):
        fo.write("%-28s" % c + "".join("%22.1f" % (max(acc[n][c]) / (1024 * 500) if acc[n].get(c) else float("nan")) for n in names) + "\n")
print(open(out + "/mix_ab.txt").read())
PY
