for r in 1 2; do
python tools/ssim_time.py
FDGS_LIB=tools/ab/libfdgs_g4.so python tools/ssim_time.py
done
