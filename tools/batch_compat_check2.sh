#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/chk
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dose.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "compat or dose or executable or fullsize or ranecu or math" 2>&1 | tail -4
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
timeout 300 python bench.py $B > /dev/null 2>&1
for wl in cirs thorax; do timeout 200 python bench.py $B --workload $wl >/dev/null 2>&1; done
rm -f gpurun_out/chk/sweep6.txt
for wl in catphan cirs thorax; do
  echo "== $wl" | tee -a gpurun_out/chk/sweep6.txt
  H=1e8 timeout 600 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,-1" "16,4,24,4" "24,4,24,4" "32,4,24,4" "32,4,16,4" "40,4,20,4" "48,4,16,4" "40,8,20,8" "40,4,20,8" "40,4,32,4" 2>&1 | tail -11 | tee -a gpurun_out/chk/sweep6.txt
done
