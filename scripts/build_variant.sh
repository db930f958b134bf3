#!/bin/bash
# scratch: build libgraphlily_hip.so of a git revision into scripts/_variants/<name>.so (git-ignored, travels with gpurun)
# usage: bash scripts/build_variant.sh HEAD head
set -e
REV=${1:-HEAD}; NAME=${2:-head}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" graphlily_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/scripts/_variants"
cd "$TMP/graphlily_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fopenmp -I../../include -I. -Wall -Wno-unused-function \
    -x hip gl_runtime.hip gl_spmv.hip gl_spmv_bool.hip gl_spmspv.hip gl_apply.hip gl_npz.cpp -shared -o "$ROOT/scripts/_variants/$NAME.so" -lz
rm -rf "$TMP"
ls -la "$ROOT/scripts/_variants/$NAME.so"
