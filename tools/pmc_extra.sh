#!/bin/bash
# Latency-side PMC counters of the bench kernel (in-flight levels / instruction counts = mean latencies; instruction fetch).
# Usage (GPU box, repo root): tools/pmc_extra.sh <out_dir>
set -u
OUT=${1:-gpurun_out/pmcx}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat"
i=0
for set in \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
WL=catphan; case "$ARGS" in *"--workload cirs"*) WL=cirs;; *"--workload thorax"*) WL=thorax;; esac
python3 tools/pmc_summary.py "$OUT" $WL > "$OUT/summary.json"
cat "$OUT/summary.json"
