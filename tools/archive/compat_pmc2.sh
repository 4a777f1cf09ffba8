#!/bin/bash
# Wait / latency counters of the COMPAT kernel on a bench workload: compat_pmc2.sh <out dir> <workload dir>
set -u
OUT=$1; WD=$2; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pass1 -- python3 tools/compat_one.py $WD 2e7 > $OUT/pass1.txt 2> $OUT/pass1.err
python3 - $OUT <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/pass1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "track_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print(acc)
if acc:
    wc = acc["SQ_WAVE_CYCLES"]
    print("waves waiting (any reason) %.3f of their cycles, waiting for an instruction to be issued %.3f, issuing %.3f" % (acc["SQ_WAIT_ANY"] / wc, acc["SQ_WAIT_INST_ANY"] / wc, acc["SQ_ACTIVE_INST_ANY"] / wc))
    print("vector memory reads per history %.2f, mean latency %.0f cycles; scalar loads per history %.2f" % (acc["SQ_INSTS_VMEM_RD"] / 2e7 , acc["SQ_INST_LEVEL_VMEM"] / max(acc["SQ_INSTS_VMEM_RD"], 1), acc["SQ_INSTS_SMEM"] / 2e7))
PY
