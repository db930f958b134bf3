#!/bin/bash
# scratch: build libgraphlily_hip.so of a git revision into scripts/_variants/<name>.so (git-ignored, travels with gpurun)
# usage: bash scripts/build_variant.sh HEAD head
set -e
REV=${1:-HEAD}; NAME=${2:-head}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" graphlily_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/scripts/_variants" "$TMP/graphlily_amd/lib"
make -s -C "$TMP/graphlily_amd/csrc"
cp "$TMP/graphlily_amd/lib/libgraphlily_hip.so" "$ROOT/scripts/_variants/$NAME.so"
rm -rf "$TMP"
ls -la "$ROOT/scripts/_variants/$NAME.so"
