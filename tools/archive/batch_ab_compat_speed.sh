#!/bin/bash
# COMPAT speed of every engine build under build/ab/*.so on the three bench workloads (one box)
cd /root/repo; mkdir -p gpurun_out/ab
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
timeout 300 python bench.py $B > /dev/null 2>&1
for wl in cirs thorax; do timeout 200 python bench.py $B --workload $wl >/dev/null 2>&1; done
for rep in 1; do for lib in build/ab/*.so; do for wl in catphan cirs thorax; do
  echo -n "$(basename $lib) $wl " | tee -a gpurun_out/ab/compat_speed.txt
  MCGPU_AMD_LIB=$PWD/$lib H=1e8 timeout 100 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,-1" 2>&1 | tail -1 | tee -a gpurun_out/ab/compat_speed.txt
done; done; done
