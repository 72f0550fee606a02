cd /tmp; export TMPDIR=/tmp
for v in bin_probe bin_probe_v4; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d /tmp/p_$v -o a -- $GRAFT_REPO_ROOT/tools/probe/$v 300000 1352 1014 18 0 q > /tmp/p_$v.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for fn in glob.glob("/tmp/p_$v/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("$v")
for k in sorted(tot):
    if "tile_" not in k: continue
    print(" ", k, {c: int(tot[k][c] / max(len(n[(k, c)]), 1)) for c in sorted(tot[k])})
PY
done
