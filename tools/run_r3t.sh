set -x
mkdir -p gpurun_out/r3t
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py $B > gpurun_out/r3t/$name.json 2>/dev/null
}
run base_1 X=1
run bwd4 FDGS_BWD_PERSIST=4
run bwd5 FDGS_BWD_PERSIST=5
run bwd3 FDGS_BWD_PERSIST=3
run bwd5_fwd5 FDGS_BWD_PERSIST=5 FDGS_FWD_PERSIST=5
run bwd4_fwd4 FDGS_BWD_PERSIST=4 FDGS_FWD_PERSIST=4
run fwd5 FDGS_FWD_PERSIST=5
run base_2 X=1
FDGS_BWD_PERSIST=4 FDGS_FWD_PERSIST=4 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward or c3_full or tile_order" > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3t/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['forward_ms'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('blend')})
PY
