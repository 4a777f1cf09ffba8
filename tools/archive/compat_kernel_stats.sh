#!/bin/bash
# rocprofv3 --kernel-trace --stats of the COMPAT kernel on the three bench workloads (1e8 histories per launch, reference launch shape)
set -u
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; OUT=gpurun_out/compat_kstats; mkdir -p $OUT
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
for wl in catphan cirs thorax; do
  [ -d /tmp/mcgpu_bench_${wl}_512_894 ] || timeout 300 python bench.py $B --workload $wl > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -- python3 tools/compat_one.py /tmp/mcgpu_bench_${wl}_512_894 1e8 > $OUT/$wl.txt 2> $OUT/$wl.err
  echo "== $wl: $(cat $OUT/$wl.txt | tail -1)"
  grep "track_kernel" $(find $OUT/$wl -name "*kernel_stats.csv" | head -1) | cut -c1-200
done
