mkdir -p gpurun_out/r3H
python -c "import torch; print(torch.cuda.Stream.priority_range())"
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
for r in 1 2; do
for cfg in "0 0" "0 -1" "-1 0" "0 -2"; do
set -- $cfg
FDGS_PRIO_F=$1 FDGS_PRIO_B=$2 python bench.py $B > gpurun_out/r3H/f$1_b$2_$r.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3H/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'])
PY
