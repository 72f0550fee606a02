set -x
mkdir -p gpurun_out/r3j
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r3j/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
FDGS_BIN_HIST=0 python bench.py $B > gpurun_out/r3j/step_nohist_$r.json 2>/dev/null
python bench.py $B > gpurun_out/r3j/step_hist_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], d['host_ms_per_view'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('tile')})
PY
