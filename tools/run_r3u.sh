set -x
mkdir -p gpurun_out/r3u
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2; do
for v in head f8 g4 b8; do
FDGS_LIB=tools/ab/libfdgs_$v.so python bench.py $B > gpurun_out/r3u/step_${v}_$r.json 2>/dev/null
done
done
FDGS_LIB=tools/ab/libfdgs_b8.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward or c3_full" > gpurun_out/r3u/pytest_b8.log 2>&1; echo "b8 pytest rc $?"
FDGS_LIB=tools/ab/libfdgs_f8.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward or c3_full" > gpurun_out/r3u/pytest_f8.log 2>&1; echo "f8 pytest rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3u/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('blend')})
PY
