# kernel trace of the two-stream step + where the wall time goes (tools/step_timeline.py); run through gpurun
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py --steps 30 --warmup 10 --min-timed-ms 0 --cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0 --reflists-steps 0 --clustered-steps 0 --axis-steps 0 --c5-steps 0 "$@" > $OUT/bench.log 2>&1
cd $REPO
CSV=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
echo "trace: $CSV"
python tools/step_timeline.py $CSV 0.3 0.55 | tee $OUT/step_timeline.txt
python tools/timeline.py $CSV 0.4 44 | tee $OUT/timeline_slice.txt
rm -f $CSV
