#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03u
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
timeout 300 python bench.py $B > /dev/null 2>&1
for wl in cirs thorax; do timeout 200 python bench.py $B --workload $wl >/dev/null 2>&1; done
for wl in catphan cirs thorax; do
  echo "== $wl" | tee -a gpurun_out/r03u/compat_sweep2.txt
  H=1e8 timeout 600 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,8" "-1,-1,-1,1" "-1,-1,-1,4" "-1,-1,-1,16" "-1,-1,-1,32" "40,4,12,8" "56,4,12,8" "48,4,8,8" "48,4,20,8" "48,8,12,8" "48,2,12,8" "24,4,16,8" "16,4,16,8" "20,4,8,8" "20,4,24,8" 2>&1 | tail -16 | tee -a gpurun_out/r03u/compat_sweep2.txt
done
bash tools/compat_pmc.sh gpurun_out/r03u/compat_pmc_thorax /tmp/mcgpu_bench_thorax_512_894 2>&1 | tail -4
bash tools/compat_pmc.sh gpurun_out/r03u/compat_pmc_catphan /tmp/mcgpu_bench_catphan_512_894 2>&1 | tail -4
