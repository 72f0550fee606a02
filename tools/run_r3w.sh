mkdir -p gpurun_out/r3K
python -m pytest tests -m gpu -x -q -k "parity or golden or tile_cull or api" 2>&1 | tail -2
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2 3; do
python bench.py $B > gpurun_out/r3K/new_$r.json 2>/dev/null
FDGS_LIB=tools/ab/libfdgs_old.so python bench.py $B > gpurun_out/r3K/old_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3K/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], d['stages']['blend_fwd']['ms'])
PY
