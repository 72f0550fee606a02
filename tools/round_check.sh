#!/usr/bin/env bash
# One GPU-box call (through gpurun) for the end of a round: the GPU test suite, the bench line in the driver's form and in the
# A/B forms the docs quote, the profile summaries (tools/collect_profiles.sh) and the step timeline.  Everything lands under
# gpurun_out/<tag>/ and gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r00}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err
L="--dropin-steps 0 --spatial-order-steps 0 --reflists-steps 0 --clustered-steps 0 --host-cost-steps 0 --cpu-samples 0 --axis-steps 0 --c5-steps 0"
python bench.py --steps 20 --warmup 5 --no-lazy $L > $O/bench_nolazy.json 2>> $O/err.log
python bench.py --steps 20 --warmup 5 --no-overlap $L > $O/bench_nooverlap.json 2>> $O/err.log
python bench.py --steps 10 --warmup 3 --workload C5 $L > $O/bench_C5.json 2>> $O/err.log
bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
bash tools/collect_profiles.sh $TAG C5 > $O/collect_C5.log 2>&1
bash tools/run_timeline.sh > $O/timeline.log 2>&1
cp gpurun_out/timeline/step_timeline.txt $O/ 2>/dev/null
ls $O
