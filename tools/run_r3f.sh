set -x
mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_api.py tests/test_gpu_train.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/r3f/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
python bench.py $B --no-batch-views > gpurun_out/r3f/step_perview_$r.json 2>/dev/null
python bench.py $B > gpurun_out/r3f/step_batched_$r.json 2>/dev/null
done
python bench.py $B --no-overlap --steps 20 --no-batch-views > gpurun_out/r3f/single_perview.json 2>/dev/null
python bench.py $B --no-overlap --steps 20 > gpurun_out/r3f/single_batched.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items()})
PY
