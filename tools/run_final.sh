set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py > gpurun_out/final/bench_default2.json 2>/dev/null
bash tools/collect_profiles.sh r03 > gpurun_out/final/collect.log 2>&1; tail -3 gpurun_out/final/collect.log
python bench.py --workload C5 --steps 20 --warmup 5 --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final/bench_C5.json 2>/dev/null
python bench.py --views-per-step 1 --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final/bench_v1.json 2>/dev/null
python bench.py --no-overlap --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final/bench_nooverlap.json 2>/dev/null
python bench.py --no-tile-cull --dropin-steps 0 > gpurun_out/final/bench_reflists.json 2>/dev/null
