mkdir -p gpurun_out/r3J
FDGS_LOSS_STREAM=1 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "pipeline" 2>&1 | tail -2
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2 3; do
python bench.py $B > gpurun_out/r3J/base_$r.json 2>/dev/null
FDGS_LOSS_STREAM=1 python bench.py $B > gpurun_out/r3J/ls_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3J/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['ms_per_step_median'])
PY
