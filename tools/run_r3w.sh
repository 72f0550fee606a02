mkdir -p gpurun_out/r3F
python -m pytest tests -m gpu -x -q -k "backward or parity or api or golden or tile_cull or analytic or densif or harness" 2>&1 | tail -3
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for w in 0 512 1172 1536; do
FDGS_PRE_BWD_WGS=$w python bench.py $B > gpurun_out/r3F/r_w$w.json 2>/dev/null
FDGS_PRE_BWD_WGS=$w python bench.py $B --spatial-order > gpurun_out/r3F/m_w$w.json 2>/dev/null
done
FDGS_LIB=tools/ab/libfdgs_cull.so python bench.py $B > gpurun_out/r3F/r_old.json 2>/dev/null
FDGS_LIB=tools/ab/libfdgs_cull.so python bench.py $B --spatial-order > gpurun_out/r3F/m_old.json 2>/dev/null
python bench.py $B --workload C5 --steps 10 --warmup 3 > gpurun_out/r3F/c5_new.json 2>/dev/null
FDGS_LIB=tools/ab/libfdgs_cull.so python bench.py $B --workload C5 --steps 10 --warmup 3 > gpurun_out/r3F/c5_old.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3F/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'bwd' in k})
PY
