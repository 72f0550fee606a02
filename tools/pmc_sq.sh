#!/usr/bin/env bash
# SQ counter passes for the blend kernels (run on the GPU box).  Two passes of <= 8 SQ counters each.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_sq
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o a -- ${FDGS_PMC_CMD:-python $REPO/tools/quick_time.py ${1:-C3} 3} > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/p2 -o b -- ${FDGS_PMC_CMD:-python $REPO/tools/quick_time.py ${1:-C3} 3} > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p3 -o c -- ${FDGS_PMC_CMD:-python $REPO/tools/quick_time.py ${1:-C3} 3} > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for fn in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("blend", "preprocess", "ssim", "sh_bwd", "tile_")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
# the kernels' durations in the pass that counted SQ_ACTIVE_INST_VALU (p2): the issue fraction is counted cycles / THESE durations
dur = collections.defaultdict(list)
for fn in glob.glob("$OUT/p2/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(tot):
    print(k)
    if dur.get(k):
        print("   %-26s %16.0f per launch" % ("_AVG_DURATION_NS", sum(dur[k]) / len(dur[k])))
    for c in sorted(tot[k]):
        print("   %-26s %16.0f per launch" % (c, tot[k][c] / max(len(n[(k, c)]), 1)))
PY
