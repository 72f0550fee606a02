set -x
mkdir -p gpurun_out/r3i
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
for v in head sh3 pb4; do
FDGS_LIB=tools/ab/libfdgs_$v.so python bench.py $B > gpurun_out/r3i/step_${v}_$r.json 2>/dev/null
done
done
FDGS_LIB=tools/ab/libfdgs_sh3.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward or c3_full" > gpurun_out/r3i/pytest_sh3.log 2>&1; echo "sh3 pytest rc $?"
FDGS_LIB=tools/ab/libfdgs_pb4.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward or c3_full" > gpurun_out/r3i/pytest_pb4.log 2>&1; echo "pb4 pytest rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items() if k in ('sh_bwd','preprocess_bwd','blend_bwd')})
PY
