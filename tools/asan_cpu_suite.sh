#!/bin/bash
# Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on the pool).
# Builds a second copy of the engine library with the host side instrumented (device code untouched: -fno-gpu-sanitize) in a
# scratch directory and runs the CPU test suite against it.  Usage: bash tools/asan_cpu_suite.sh [tsan]  (prints the report count;
# "tsan": ThreadSanitizer instead, for the threaded parser and writers)
set -e
SAN="address,undefined"; RTNAME=asan; SKIP=""
# torch's own gloo worker threads are not instrumented and report races among themselves: the two-rank test is left out under tsan
[ "$1" = tsan ] && { SAN=thread; RTNAME=tsan; SKIP="--ignore=tests/test_distributed_cpu.py"; }
ROOT=$(cd "$(dirname "$0")/.." && pwd); W=${TMPDIR:-/tmp}/mcgpu_asan_build
rm -rf "$W"; mkdir -p "$W/pkg"; cp -r "$ROOT/4d-cbct-mc_amd/csrc" "$W/pkg/"; cp -r "$ROOT/include" "$W/"
cd "$W/pkg/csrc"; make clean > /dev/null 2>&1 || true
sed -i 's/^CXXFLAGS := -O3/CXXFLAGS := -O1 -g -fsanitize='$SAN' -fno-gpu-sanitize -fno-omit-frame-pointer/' Makefile
sed -i 's#-shared -o \$@ \$(OBJS)#-shared -fsanitize='$SAN' -o $@ $(OBJS)#' Makefile
make -j8 ../libmcgpu_amd.so > "$W/build.log" 2>&1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.$RTNAME-x86_64.so)
cd "$ROOT"
MCGPU_AMD_LIB="$W/pkg/libmcgpu_amd.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/ $SKIP -x -q -m "not gpu" -p no:cacheprovider -s > "$W/run.log" 2>&1 || true
tail -1 "$W/run.log"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer\|ThreadSanitizer' "$W/run.log")"
