#!/bin/bash
# COMPAT kernel with two batches per lane: parity first, then the threshold sweep on the three bench workloads
cd /root/repo; mkdir -p gpurun_out/r03u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dose.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "compat or dose or executable or fullsize" 2>&1 | tail -8 | tee gpurun_out/r03u/compat_parity.txt
timeout 300 python bench.py --steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1   # builds the catphan workload dir
for wl in cirs thorax; do timeout 200 python bench.py --steps 2 --warmup 1 --workload $wl --no-workloads --no-cpu-baseline --no-end-to-end --no-compat >/dev/null 2>&1; done
ls -d /tmp/mcgpu_bench_*
for wl in catphan cirs thorax; do
  echo "== $wl" | tee -a gpurun_out/r03u/compat_sweep.txt
  H=1e8 timeout 600 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1" "48,4,12" "64,8,24" "80,8,32" "96,8,32" "96,16,48" "112,16,64" "64,4,16" "32,4,12" 2>&1 | tail -12 | tee -a gpurun_out/r03u/compat_sweep.txt
done
