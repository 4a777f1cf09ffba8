#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean value per track_kernel dispatch, per counter."""
import csv, glob, json, sys
out = {}
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "track_" not in row["Kernel_Name"] or "kernel" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, d in acc.items():
        v = list(d.values())
        out[name] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
print(json.dumps(out, indent=1))
