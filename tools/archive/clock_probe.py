#!/usr/bin/env python3
"""Dev measurement (GPU): shader clock the FAST kernel actually runs at (diagnostic build: per-wave s_memtime cycles over
s_memrealtime ticks of 100 MHz), per workload.  Usage: python tools/clock_probe.py [catphan thorax ...]"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
__import__("os").environ.setdefault("MCGPU_AMD_LIB", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "4d-cbct-mc_amd", "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import bench, cases
eng = cases.pkg.engine
NS = 32
NT = NS + 3 * 16384
for wl in (sys.argv[1:] or ["catphan"]):
    wd = Path(f"/tmp/mcgpu_wl_{wl}"); wd.mkdir(exist_ok=True)
    inp = wd / "input.in"
    if not inp.exists():
        inp = bench.build_workload(wd, wl, int(1e8), 894, eng)
    with eng.create(inp, device=0) as ctx:
        for rep in range(3):
            out = (C.c_ulonglong * NT)()
            ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 1)
            _, secs, done = ctx.run_projection(7, int(1e8), mode="stats", seed=1)
            ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 0)
        a = np.frombuffer(out, dtype=np.uint64)[NS:].reshape(-1, 3)
        a = a[a[:, 1] != 0]
        cyc = ((a[:, 0] >> np.uint64(36)).astype(np.float64)) * 256.0
        ticks = (a[:, 2].astype(np.int64) - a[:, 1].astype(np.int64)).astype(np.float64)
        ok = ticks > 1000
        mhz = cyc[ok] / ticks[ok] * 100.0
        print(f"{wl}: stats-build kernel {secs*1e3:.2f} ms; clock over {ok.sum()} waves: median {np.median(mhz):.0f} MHz, p5 {np.percentile(mhz,5):.0f}, p95 {np.percentile(mhz,95):.0f}")
