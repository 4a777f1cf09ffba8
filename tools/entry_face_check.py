#!/usr/bin/env python3
"""Dev measurement (GPU): the entry-face shell at higher statistics than the bench line affords -- FAST against COMPAT, primary energy per
history at an oblique thorax projection (bench_legs/checks.py: entry_face_deficit).  usage: python tools/entry_face_check.py [runs [projection]]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, cases
eng = cases.pkg.engine
wd = Path("/tmp/mcgpu_bench_thorax_512_894")
if not (wd / "input.in").exists():
    wd.mkdir(parents=True, exist_ok=True)
    bench.build_workload(wd, "thorax", int(1e8), 894, eng)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = int(sys.argv[2]) if len(sys.argv) > 2 else 600
with eng.create(wd / "input.in", device=0) as ctx:
    print(json.dumps(bench.entry_face_deficit(ctx, runs=runs, fast_histories=10_000_000_000, compat_histories=5_000_000_000, projection=p), indent=1))
