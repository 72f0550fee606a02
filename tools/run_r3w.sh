mkdir -p gpurun_out/r3B
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for pad in 0 2048; do
for sm in 0 1 2 4; do
FDGS_BWD_LDS_PAD=$pad FDGS_BIN_SMALL=$sm python bench.py $B > gpurun_out/r3B/p${pad}_s${sm}.json 2>/dev/null
done
done
FDGS_BWD_LDS_PAD=4096 FDGS_BIN_SMALL=4 python bench.py $B > gpurun_out/r3B/p4096_s4.json 2>/dev/null
FDGS_BWD_LDS_PAD=4096 FDGS_BIN_SMALL=1 python bench.py $B > gpurun_out/r3B/p4096_s1.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3B/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    st=d['stages']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:st[k]['ms'] for k in ('tile_count','tile_scatter','blend_bwd')})
PY
