#!/usr/bin/env bash
# Builds libfdgs.so from another git revision's csrc/ into tools/ab/libfdgs_<name>.so (git-ignored; travels to the GPU
# box with gpurun), for A/B timing of kernel variants in ONE gpurun call (box-to-box variance is larger than most kernel
# changes):   tools/ab_build.sh HEAD base  &&  gpurun -- 'FDGS_LIB=tools/ab/libfdgs_base.so python bench.py ...; python bench.py ...'
# The library must be ABI-compatible with the working tree's _capi.py.
set -euo pipefail
REF=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REF" 4d-gaussian-splatting_amd/csrc include | tar -x -C "$TMP"
bash "$TMP/4d-gaussian-splatting_amd/csrc/build.sh" > /dev/null
mkdir -p "$ROOT/tools/ab"
cp "$TMP/4d-gaussian-splatting_amd/csrc/libfdgs.so" "$ROOT/tools/ab/libfdgs_$NAME.so"
rm -rf "$TMP"
echo "$ROOT/tools/ab/libfdgs_$NAME.so"
