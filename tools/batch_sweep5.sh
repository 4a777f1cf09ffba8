#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/chk
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
timeout 300 python bench.py $B > /dev/null 2>&1
for wl in cirs thorax; do timeout 200 python bench.py $B --workload $wl >/dev/null 2>&1; done
for wl in catphan cirs thorax; do
  echo "== $wl" | tee -a gpurun_out/chk/sweep5.txt
  H=1e8 timeout 600 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,-1" "16,4,24,4" "24,4,24,4" "32,4,16,4" "36,4,12,4" "40,4,12,4" "44,4,12,4" "52,4,12,4" "40,4,20,4" "40,8,16,8" "32,4,24,2" 2>&1 | tail -12 | tee -a gpurun_out/chk/sweep5.txt
done
bash tools/compat_pmc.sh gpurun_out/chk/pmc_thorax /tmp/mcgpu_bench_thorax_512_894 2>&1 | tail -3
