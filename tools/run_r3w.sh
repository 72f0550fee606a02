mkdir -p gpurun_out/r3A
python -m pytest tests -m gpu -x -q -k "sh or backward or api or golden" > gpurun_out/r3A/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3A/pytest.log
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for w in 0 1024 1536 3072 4688; do
FDGS_SH_BWD_WAVES=$w python bench.py $B > gpurun_out/r3A/r_w$w.json 2>/dev/null
FDGS_SH_BWD_WAVES=$w python bench.py $B --spatial-order > gpurun_out/r3A/m_w$w.json 2>/dev/null
done
FDGS_LIB=tools/ab/libfdgs_g4.so python bench.py $B > gpurun_out/r3A/r_old.json 2>/dev/null
FDGS_LIB=tools/ab/libfdgs_g4.so python bench.py $B --spatial-order > gpurun_out/r3A/m_old.json 2>/dev/null
for w in 0 4096 8192 16384; do
FDGS_SH_BWD_WAVES=$w python bench.py $B --workload C5 --steps 10 --warmup 3 > gpurun_out/r3A/c5_w$w.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3A/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('sh')})
PY
