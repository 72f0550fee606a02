python -m pytest tests -m gpu -x -q -k "ssim or loss" 2>&1 | tail -2
for r in 1 2; do
for v in old new h48; do
if [ $v = new ]; then python tools/ssim_time.py; else FDGS_LIB=tools/ab/libfdgs_$v.so python tools/ssim_time.py; fi
done
done
