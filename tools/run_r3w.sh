mkdir -p gpurun_out/r3I
FDGS_SORT_SPLIT=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning or c3_full or run_ahead" 2>&1 | tail -2
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2; do
python bench.py $B > gpurun_out/r3I/base_$r.json 2>/dev/null
FDGS_SORT_SPLIT=1 python bench.py $B > gpurun_out/r3I/split_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3I/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], d['stages']['tile_sort']['ms'])
PY
