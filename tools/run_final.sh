mkdir -p gpurun_out/final3
python -m pytest tests -m gpu -x -q > gpurun_out/final3/pytest_gpu.log 2>&1; tail -2 gpurun_out/final3/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh r03 > gpurun_out/final3/collect.log 2>&1
python bench.py > gpurun_out/final3/bench_default.json 2>/dev/null
python bench.py --workload C5 --steps 20 --warmup 5 --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final3/bench_C5.json 2>/dev/null
python bench.py --views-per-step 1 --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final3/bench_v1.json 2>/dev/null
python bench.py --no-overlap --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final3/bench_nooverlap.json 2>/dev/null
python bench.py --no-tile-cull --dropin-steps 0 > gpurun_out/final3/bench_reflists.json 2>/dev/null
