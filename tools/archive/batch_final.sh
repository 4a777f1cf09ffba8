#!/bin/bash
# End-of-round check + collection (GPU box): GPU suite, smoke, collect_round.sh, COMPAT sweep / PMC / counters.  Usage: bash tools/batch_final.sh <tag>
TAG=${1:-rXX}
cd /root/repo; mkdir -p gpurun_out/${TAG}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/${TAG}/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/${TAG}/smoke.txt
bash tools/collect_round.sh ${TAG} 2>&1 | tail -30
for wl in catphan cirs thorax; do
  echo "== $wl" | tee -a gpurun_out/${TAG}/compat_sweep4.txt
  H=1e8 timeout 400 python tools/compat_sweep.py /tmp/mcgpu_bench_${wl}_512_894 "-1,-1,-1,-1" "24,4,24,4" "32,4,24,4" "40,4,20,4" "48,4,16,4" "56,4,16,4" 2>&1 | tail -7 | tee -a gpurun_out/${TAG}/compat_sweep4.txt
done
bash tools/compat_pmc.sh gpurun_out/${TAG}/compat_pmc_thorax3 /tmp/mcgpu_bench_thorax_512_894 2>&1 | tail -4 | tee gpurun_out/${TAG}/compat_pmc_thorax3.txt
bash tools/compat_stats.sh 2>&1 | tail -3 | tee gpurun_out/${TAG}/compat_stats.txt
