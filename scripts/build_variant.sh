#!/bin/bash
# scratch: build libgraphlily_hip.so of a git revision (or WORK = the working tree) into scripts/_variants/<name>.so
# (git-ignored, travels with gpurun); extra compiler flags after the name
# usage: bash scripts/build_variant.sh HEAD head        bash scripts/build_variant.sh WORK plain -DGL_STREAM_PLAIN
#   ONLY="gl_spmv.hip" (WORK builds): start from the in-tree objects and recompile only these units with the extra flags
#   (a flag that only one translation unit looks at: minutes saved per variant)
set -e
REV=${1:-HEAD}; NAME=${2:-head}; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
if [ "$REV" = WORK ]; then
  mkdir -p "$TMP/graphlily_amd" && cp -r "$ROOT/graphlily_amd/csrc" "$TMP/graphlily_amd/csrc" && cp -r "$ROOT/include" "$TMP/include"
else
  git -C "$ROOT" archive "$REV" graphlily_amd/csrc include | tar -x -C "$TMP"
fi
mkdir -p "$ROOT/scripts/_variants" "$TMP/graphlily_amd/lib"
if [ "$REV" = WORK ] && [ -n "$ONLY" ] && [ -d "$ROOT/graphlily_amd/lib/obj" ]; then
  cp -r "$ROOT/graphlily_amd/lib/obj" "$TMP/graphlily_amd/lib/obj"
  touch "$TMP"/graphlily_amd/lib/obj/*.o
  for f in $ONLY; do rm -f "$TMP/graphlily_amd/lib/obj/${f%.*}.o"; done
fi
make -s -j4 -C "$TMP/graphlily_amd/csrc" EXTRA="$*"
cp "$TMP/graphlily_amd/lib/libgraphlily_hip.so" "$ROOT/scripts/_variants/$NAME.so"
rm -rf "$TMP"
ls -la "$ROOT/scripts/_variants/$NAME.so"
