#!/usr/bin/env bash
# VGPR / SGPR / spill counts of every kernel of one translation unit: tools/kregs.sh sh_bwd [extra hipcc flags]
set -euo pipefail
SRC=$1; shift || true
CS=$(cd "$(dirname "$0")/../4d-gaussian-splatting_amd/csrc" && pwd)
TMP=$(mktemp -d)
EXTRA=""
case $SRC in preprocess_fwd|preprocess_bwd|sh_bwd) EXTRA="-ffp-contract=off -fno-slp-vectorize";; knn) EXTRA="-ffp-contract=off";; ssim) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $EXTRA "$@" -S --cuda-device-only -o $TMP/k.s $CS/$SRC.hip
python3 - $TMP/k.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [None, '?'])[1]
    print("%-70s vgpr %4s spill %3s  sgpr %4s spill %3s  lds %6s" % (name[:70], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count'), g('group_segment_fixed_size')))
PY
rm -rf $TMP
