#!/usr/bin/env bash
# Builds libfdgs.so (C ABI in include/fdgs.h) for gfx950 with hipcc.  In-tree, no GPU needed.
#   ./build.sh            incremental (per-TU objects under build/)
#   ./build.sh clean      remove objects and the library
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${FDGS_ARCH:-gfx950}   # gfx950 only: colour_batch_kernel takes ~75 KB of static LDS (160 KB per CU here; a 64 KB-LDS target does not build)
OUT=libfdgs.so
if [[ "${1:-}" == "clean" ]]; then rm -rf build "$OUT"; exit 0; fi
mkdir -p build
COMMON="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
# TUs whose float results feed integers (radius / tile rect / depth key bits) or must track the
# oracle's operation order are built with FP contraction off (no FMA fusion).
# -fno-slp-vectorize on the per-Gaussian kernels (round 6): the vectorizer turns their scalar fp32 chains into packed operations on register
# PAIRS, with one or two moves per packed operation to line the operands up -- preprocess_fwd 235 -> 173 VGPRs (its colour half 214 -> 153:
# three waves per SIMD), preprocess_bwd 132 -> 97 (five waves), sh_bwd 222 -> 211 and 6 % fewer instructions; the same IEEE operations one at
# a time, bit-identical results (the whole GPU suite); step +0.4 % at C3, +0.9 % at C5, forward-only -3 %.
# ssim: the SLP vectorizer pairs the scalar third channel of the backward's windows into packed multiplies that each need four register
# moves to line their operands up (595 -> 568 instructions per thread without it; the forward is packed by hand either way)
declare -A EXTRA=( [preprocess_fwd]="-ffp-contract=off -fno-slp-vectorize" [preprocess_bwd]="-ffp-contract=off -fno-slp-vectorize" [sh_bwd]="-ffp-contract=off -fno-slp-vectorize" [knn]="-ffp-contract=off" [ssim]="-fno-slp-vectorize" )
OBJS=()
PIDS=()
for src in preprocess_fwd tilebin radix_sort blend_fwd blend_bwd preprocess_bwd sh_bwd ssim adam densify knn capi; do
  obj=build/$src.o
  OBJS+=("$obj")
  if [[ ! -f $obj || $src.hip -nt $obj || fdgs_common.h -nt $obj || blend_common.h -nt $obj || fdgs_math.h -nt $obj || ../../include/fdgs.h -nt $obj ]]; then
    $HIPCC $COMMON ${EXTRA[$src]:-} ${FDGS_EXTRA_FLAGS:-} -c $src.hip -o $obj &
    PIDS+=($!)
  fi
done
for p in "${PIDS[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
$HIPCC --offload-arch=$ARCH -shared -fPIC -o $OUT "${OBJS[@]}"
echo "$(pwd)/$OUT"
