mkdir -p gpurun_out/final2
python bench.py > gpurun_out/final2/bench_default.json 2>/dev/null
python bench.py > gpurun_out/final2/bench_default2.json 2>/dev/null
python bench.py --workload C5 --steps 20 --warmup 5 --dropin-steps 0 --spatial-order-steps 0 > gpurun_out/final2/bench_C5.json 2>/dev/null
