#!/bin/bash
# Builds the diagnostic variant of the COMPAT kernel (wave-level counters, -DMC_COMPAT_STATS) as build/ab/compat_stats.so next to the
# product library and prints the time breakdown on the bench workloads.  Usage (GPU box, repo root): bash tools/compat_stats.sh
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; mkdir -p build/ab gpurun_out
W=$(mktemp -d); mkdir -p "$W/pkg"; cp -rp 4d-cbct-mc_amd/csrc "$W/pkg/csrc"; cp -rp include "$W/include"
( cd "$W/pkg/csrc" && rm -f track_compat.o && sed -i 's/-ffp-contract=off -c \$< -o \$@/-ffp-contract=off -DMC_COMPAT_STATS=1 -c $< -o $@/' Makefile && make -j8 ../libmcgpu_amd.so > "$W/build.log" 2>&1 ) || { tail -5 "$W/build.log"; exit 1; }
cp "$W/pkg/libmcgpu_amd.so" build/ab/compat_stats.so
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
for wl in catphan cirs thorax; do
  [ -d /tmp/mcgpu_bench_${wl}_512_894 ] || timeout 300 python bench.py $B --workload $wl > /dev/null 2>&1
  echo -n "$wl " ; MCGPU_AMD_LIB=$ROOT/build/ab/compat_stats.so MCGPU_COMPAT_STATS=1 timeout 200 python tools/compat_stats.py /tmp/mcgpu_bench_${wl}_512_894 2e7 2>&1 | tail -1
done | tee gpurun_out/compat_stats.txt
