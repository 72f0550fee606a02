set -x
mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_api.py tests/test_gpu_train.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/r3g/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
python bench.py $B --no-tails-on-f > gpurun_out/r3g/step_perview_$r.json 2>/dev/null
python bench.py $B > gpurun_out/r3g/step_tails_$r.json 2>/dev/null
python bench.py $B --batch-views > gpurun_out/r3g/step_batched_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items()})
PY
