#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean value per FAST-kernel dispatch, per counter, stamped with the kernel build
(hash of the FAST kernel's sources, bench.kernel_source_hash), the workload, the kernel VARIANT the engine dispatches for it (tile
records or plain u8, scheduler: chosen at run time in model_device.cpp) and the MCGPU_* knobs of the collecting environment, so
that bench.py can refuse a summary that belongs to another build, variant or tuning.  Usage: pmc_summary.py <dir with pass*/> [workload]"""
import csv, glob, json, os, re, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
out = {}
kernels = set()
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "track_" not in row["Kernel_Name"] or "kernel" not in row["Kernel_Name"]:
            continue
        m = re.search(r"track_\w+<[^>]*>", row["Kernel_Name"])
        if m and re.search(r",\s*1>$", m.group(0)) and "track_pool_kernel" in m.group(0):
            continue  # track_fast64.hip's kernel (third template argument 1): not the production kernel this summary is about
        kernels.add(m.group(0) if m else row["Kernel_Name"])
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, d in acc.items():
        v = list(d.values())
        out[name] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
workload = sys.argv[2] if len(sys.argv) > 2 else "catphan"
out["_stamp"] = {"kernel_source_sha16": bench.kernel_source_hash(), "workload": workload,
                 "kernel_names": sorted({row for row in kernels}), "variant": bench.kernel_variant(workload), "knobs": bench.knob_environment(),
                 "histories_per_dispatch": 100000000, "collected": time.strftime("%Y-%m-%d %H:%M:%S")}
print(json.dumps(out, indent=1))
