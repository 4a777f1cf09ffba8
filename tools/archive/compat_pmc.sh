#!/bin/bash
# Lane utilisation of the COMPAT kernel on a bench workload: compat_pmc.sh <out dir> <workload dir>
set -u
OUT=$1; WD=$2; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pass1 -- python3 tools/compat_one.py $WD 2e7 > $OUT/pass1.txt 2> $OUT/pass1.err
python3 - $OUT <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/pass1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "track_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print(acc)
if acc:
    print("lane utilisation of VALU instructions: %.3f" % (acc["SQ_THREAD_CYCLES_VALU"] / (64.0 * acc["SQ_ACTIVE_INST_VALU"])))
    print("VALU instructions per history (2e7): %.1f wave-instr -> %.1f per history" % (acc["SQ_INSTS_VALU"], acc["SQ_INSTS_VALU"] / 2e7))
PY
