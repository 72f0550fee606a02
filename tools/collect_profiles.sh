#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py plus two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of the rasterizer-only loop, summarised into gpurun_out/profiles_<tag>/.
set -uo pipefail
# usage: tools/collect_profiles.sh <tag e.g. r05> [workload=C3]   (C5: the summaries get the suffix _C5)
TAG=${1:-r01}
WL=${2:-C3}
SUF=""; [[ "$WL" != "C3" ]] && SUF="_$WL"
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG$SUF
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the secondary legs are switched off: the trace is the timed step + the stage pass + the forward-only and rasterizer-only legs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --workload $WL --steps $([[ $WL == C5 ]] && echo 6 || echo 20) --warmup 5 --min-timed-ms 300 --cpu-samples 0 --host-cost-steps 0 --dropin-steps 0 --spatial-order-steps 0 --reflists-steps 0 --clustered-steps 0 --axis-steps 0 --c5-steps 0 > $OUT/bench_under_rocprof.log 2>&1
# PMC passes over tools/step_loop.py: the kernel variants of the timed step (one counter per pass: MI355X_MICROARCH.md, HBM section)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python $REPO/tools/step_loop.py $WL 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python $REPO/tools/step_loop.py $WL 3 > $OUT/pmc_write.log 2>&1
cd $REPO
FDGS_PMC_CMD="python $REPO/tools/step_loop.py $WL 2" bash tools/pmc_sq.sh > $OUT/pmc_sq_$TAG$SUF.txt 2>&1
python tools/pmc_traffic.py $OUT/stats $OUT/fetch $OUT/write $OUT/pmc_traffic_$TAG$SUF
tail -1 $OUT/bench_under_rocprof.log | cut -c1-600
# keep only the summaries (raw traces are large)
rm -rf $OUT/fetch/*/*kernel_trace.csv $OUT/write/*/*kernel_trace.csv 2>/dev/null
ls -la $OUT
