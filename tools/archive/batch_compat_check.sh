#!/bin/bash
# COMPAT kernel after a change: parity tests first, then the speed of the three bench workloads at the default thresholds
cd /root/repo; mkdir -p gpurun_out/chk
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dose.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "compat or dose or executable or fullsize or ranecu or math" 2>&1 | tail -4 | tee gpurun_out/chk/parity.txt
B="--steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat"
timeout 300 python bench.py $B > /dev/null 2>&1
for wl in cirs thorax; do timeout 200 python bench.py $B --workload $wl >/dev/null 2>&1; done
for rep in 1 2; do for wl in catphan cirs thorax; do
  echo -n "$wl " | tee -a gpurun_out/chk/speed.txt
  H=1e8 timeout 100 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,-1" 2>&1 | tail -1 | tee -a gpurun_out/chk/speed.txt
done; done
