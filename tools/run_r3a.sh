set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" 
tail -5 gpurun_out/r3a/pytest.log
python bench.py > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err; echo "bench rc $?"
python bench.py --workload C3x4 --no-overlap --steps 10 --warmup 5 --cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 > gpurun_out/r3a/bench_c3x4_single.json 2> gpurun_out/r3a/bench_c3x4.err; echo "c3x4 rc $?"
python bench.py --no-overlap --steps 20 --cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 > gpurun_out/r3a/bench_c3_single.json 2>/dev/null; echo "single rc $?"
python bench.py --workload C3x4 --steps 10 --warmup 5 --cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 > gpurun_out/r3a/bench_c3x4.json 2>/dev/null; echo "c3x4 2-stream rc $?"
head -c 600 gpurun_out/r3a/bench_default.json
