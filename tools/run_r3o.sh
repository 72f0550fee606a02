set -x
mkdir -p gpurun_out/r3o
python -m pytest tests/test_gpu_api.py tests/test_gpu_train.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r3o/pytest.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r3o/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2; do
for g in 1 2 4; do
python bench.py $B --sh-group $g > gpurun_out/r3o/step_g${g}_$r.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3o/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items() if k in ('sh_bwd','preprocess_bwd')})
PY
