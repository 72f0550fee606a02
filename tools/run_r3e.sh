set -x
mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r3e/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
for v in v3 so; do
FDGS_LIB=tools/ab/libfdgs_$v.so python bench.py $B > gpurun_out/r3e/step_${v}_$r.json 2>/dev/null
done
done
for v in v3 so; do
FDGS_LIB=tools/ab/libfdgs_$v.so python bench.py $B --no-overlap --steps 20 > gpurun_out/r3e/single_$v.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['forward_ms'], d['raster_images_s'] if 'raster_images_s' in d else '', {k:v['ms'] for k,v in d['stages'].items()})
PY
