#!/usr/bin/env bash
# Builds libfdgs.so from the WORKING TREE's csrc/ with extra compiler flags into tools/ab/libfdgs_<name>.so (git-ignored; travels to the
# GPU box with gpurun) for A/B timing of compile-time variants in one gpurun call:
#   tools/ab_variant.sh w3 "-DFDGS_PRE_WAVES=3"  &&  gpurun -- 'python tools/ab_stage_env.py "" "FDGS_LIB=tools/ab/libfdgs_w3.so"'
set -euo pipefail
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/4d-gaussian-splatting_amd" "$TMP/include"
cp -r "$ROOT/4d-gaussian-splatting_amd/csrc" "$TMP/4d-gaussian-splatting_amd/csrc"
cp "$ROOT/include/fdgs.h" "$TMP/include/"
rm -rf "$TMP/4d-gaussian-splatting_amd/csrc/build" "$TMP/4d-gaussian-splatting_amd/csrc/libfdgs.so"
FDGS_EXTRA_FLAGS="$FLAGS" bash "$TMP/4d-gaussian-splatting_amd/csrc/build.sh" > /dev/null
mkdir -p "$ROOT/tools/ab"
cp "$TMP/4d-gaussian-splatting_amd/csrc/libfdgs.so" "$ROOT/tools/ab/libfdgs_$NAME.so"
rm -rf "$TMP"
echo "$ROOT/tools/ab/libfdgs_$NAME.so"
