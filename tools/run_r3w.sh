python -m pytest tests -m gpu -x -q -k "ssim or loss" 2>&1 | tail -2
for r in 1 2 3; do python tools/ssim_time.py; done
python tools/ssim_time.py 480 640
B="--cpu-samples 0 --dropin-steps 0 --host-cost-steps 0 --spatial-order-steps 0"
python bench.py $B | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
python bench.py $B | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
