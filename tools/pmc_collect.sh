#!/bin/bash
# Collects rocprofv3 PMC counters of the bench kernel in separate passes (MI355X_MICROARCH.md "rocprofv3 PMC slots":
# 8 SQ, 4 TCC slots per pass; FETCH_SIZE costs 3, WRITE_SIZE 2) and summarises them into one JSON.
# Usage (on the GPU box): tools/pmc_collect.sh <out_dir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=${*:---steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads --no-reference-arithmetic}
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
WL=catphan; case "$ARGS" in *"--workload cirs"*) WL=cirs;; *"--workload thorax_textured"*) WL=thorax_textured;; *"--workload thorax"*) WL=thorax;; esac
python3 tools/pmc_summary.py "$OUT" $WL > "$OUT/summary.json"
cat "$OUT/summary.json"
