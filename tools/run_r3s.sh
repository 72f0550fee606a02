set -x
mkdir -p gpurun_out/r3s
python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "step_pipeline or preprocess_batch" > gpurun_out/r3s/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r3s/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2; do
python bench.py $B > gpurun_out/r3s/step_random_perview_$r.json 2>/dev/null
python bench.py $B --batch-views > gpurun_out/r3s/step_random_batchcol_$r.json 2>/dev/null
python bench.py $B --spatial-order > gpurun_out/r3s/step_morton_perview_$r.json 2>/dev/null
python bench.py $B --spatial-order --batch-views > gpurun_out/r3s/step_morton_batchcol_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3s/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'])
PY
