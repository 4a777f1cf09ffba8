#!/bin/bash
# Builds a variant of the engine library under build/ab/<name>.so from a scratch copy of the sources with the given sed
# expressions applied (file:expr pairs), for A/B measurements on one GPU box (tools/ab.sh).  The tree itself is not touched.
# usage: [STATS=1] tools/build_variant.sh <name> [<file>:<sed expression> ...]      e.g.  base   or   b768 'device_model.hpp:s/kPoolBlockThreads = 1024/kPoolBlockThreads = 768/'
set -eu
cd "$(dirname "$0")/.."; ROOT=$PWD; mkdir -p build/ab
name=$1; shift
W=$(mktemp -d); mkdir -p "$W/pkg"; cp -rp 4d-cbct-mc_amd/csrc "$W/pkg/csrc"; cp -rp include "$W/include"
for pair in "$@"; do
  f=${pair%%:*}; e=${pair#*:}
  before=$(sha1sum "$W/pkg/csrc/$f")
  sed -i "$e" "$W/pkg/csrc/$f"
  [ "$before" != "$(sha1sum "$W/pkg/csrc/$f")" ] || { echo "build_variant: '$e' changed nothing in $f" >&2; exit 1; }
  touch "$W/pkg/csrc/$f"
done
# STATS=1: the diagnostic library (stats mode of the FAST kernel) instead of the product library
target=libmcgpu_amd.so; [ "${STATS:-0}" = 1 ] && target=libmcgpu_amd_stats.so
( cd "$W/pkg/csrc" && make -j8 ../$target > "$W/build.log" 2>&1 ) || { tail -20 "$W/build.log"; exit 1; }
cp "$W/pkg/$target" "build/ab/$name.so"
rm -rf "$W"
echo "built build/ab/$name.so"
