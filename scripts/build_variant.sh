#!/bin/bash
# scratch: build libgraphlily_hip.so of a git revision (or WORK = the working tree) into scripts/_variants/<name>.so
# (git-ignored, travels with gpurun); extra compiler flags after the name
# usage: bash scripts/build_variant.sh HEAD head        bash scripts/build_variant.sh WORK plain -DGL_STREAM_PLAIN
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
make -s -C "$TMP/graphlily_amd/csrc" EXTRA="$*"
cp "$TMP/graphlily_amd/lib/libgraphlily_hip.so" "$ROOT/scripts/_variants/$NAME.so"
rm -rf "$TMP"
ls -la "$ROOT/scripts/_variants/$NAME.so"
