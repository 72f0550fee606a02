set -x
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r3b/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0"
for r in 1 2; do
FDGS_TILE_ORDER=0 python bench.py $B > gpurun_out/r3b/step_noorder_$r.json 2>/dev/null
python bench.py $B > gpurun_out/r3b/step_order_$r.json 2>/dev/null
done
FDGS_TILE_ORDER=0 python bench.py $B --no-overlap --steps 20 > gpurun_out/r3b/single_noorder.json 2>/dev/null
python bench.py $B --no-overlap --steps 20 > gpurun_out/r3b/single_order.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items()})
PY
