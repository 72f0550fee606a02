set -x
mkdir -p gpurun_out/r3m
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
FDGS_PMC_CMD="python $REPO/tools/ssim_time.py 1352 1014 20" bash $REPO/tools/pmc_sq.sh > $REPO/gpurun_out/r3m/pmc_sq_ssim.txt 2>&1
cd $REPO
python tools/ssim_time.py 1352 1014 200
grep -A26 "ssim_fwd_kernel" gpurun_out/r3m/pmc_sq_ssim.txt | head -60
