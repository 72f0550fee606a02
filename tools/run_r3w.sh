python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
mkdir -p gpurun_out/r3E
python bench.py > gpurun_out/r3E/bench_default.json 2>/dev/null; tail -c 300 gpurun_out/r3E/bench_default.json
python bench.py --no-tile-cull --dropin-steps 0 > gpurun_out/r3E/bench_reflists.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_default','bench_reflists'):
    d=json.loads([l for l in open('gpurun_out/r3E/%s.json'%n) if l.startswith('{')][-1])
    print(n, d['value'], d['ms_per_step'], d['forward_ms'], d.get('raster_images_s'), d.get('spatial_order_images_s'), d['config'].get('num_rendered'), d['roofline']['frac'], d['roofline']['valu_issue_frac'], {k:v['ms'] for k,v in d['stages'].items()})
PY
