# kernel trace of a few forwards of a workload: per-launch durations of the sort instances (run through gpurun)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-C3-clustered}; MODE=${2:-cull}
OUT=$REPO/gpurun_out/sort_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $REPO/tools/quick_time.py $W 6 colour $([ "$MODE" = reflists ] && echo reflists) > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for fn in glob.glob("$OUT/t/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "tile_" in k: d[k.split("(")[0] + " grid " + r["Grid_Size_X"] + " wg " + r["Workgroup_Size_X"] + " lds " + r.get("LDS_Block_Size", "?")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()): print("%-90s n=%3d  avg %.1f us  min %.1f  max %.1f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
PY
tail -2 $OUT/log.txt
rm -rf $OUT/t
