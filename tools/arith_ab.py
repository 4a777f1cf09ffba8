#!/usr/bin/env python3
"""Dev measurement (GPU): the FAST kernel in single precision ("fast") against the same kernel with the reference's three
double-precision sub-steps ("fast64": rotate_double, GRAa, GCOa's cdt1 / costh chain), on ONE box.
  1. kernel ms of 1e8-history launches, the two modes interleaved launch by launch at the bench's projection angles;
  2. detected energy per history and scatter class, K runs per mode with different seeds: ratio fast64 / fast and its z.
Usage: python tools/arith_ab.py [workloads ...]   (LAUNCHES=12 RUNS=12 HIST=5e8)"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import cases
from bench_legs.common import build_workload
eng = cases.pkg.engine
H = 100_000_000
L = int(os.environ.get("LAUNCHES", "12")); K = int(os.environ.get("RUNS", "12")); N = int(float(os.environ.get("HIST", "5e8")))
MODES = os.environ.get("MODES", "fast,fast64").split(",")
for wl in (sys.argv[1:] or ["catphan", "cirs", "thorax", "thorax_textured"]):
    wd = Path(f"/tmp/mcgpu_bench_{wl}_512_894")
    if not (wd / "input.in").exists():
        wd.mkdir(parents=True, exist_ok=True)
        build_workload(wd, wl, H, 894, eng)
    with eng.create(wd / "input.in", device=0) as ctx:
        nz, nx = ctx.detector_shape
        image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        seed, nproj = ctx.geti("seed"), ctx.num_projections
        ms = {m: [] for m in MODES}
        for i in range(3 + L):
            for m in MODES:
                ctx.clear(image.data_ptr(), stream)
                ctx.launch((i * 149) % nproj, image.data_ptr(), H, mode=m, seed=seed, stream=stream)
                t = ctx.last_kernel_ms()
                if i >= 3:
                    ms[m].append(t)
        base = float(np.mean(ms[MODES[0]]))
        print(f"{wl}: " + "  ".join(f"{m} {np.mean(ms[m]):.3f} ms (min {np.min(ms[m]):.3f}, {H / np.mean(ms[m]) / 1e6:.2f}e9 hist/s, {np.mean(ms[m]) / base - 1:+.1%})" for m in MODES), flush=True)
        if K > 0:
            e = {m: [] for m in MODES}
            t0 = time.time()
            for k in range(K):
                for j, m in enumerate(MODES):
                    img, _, d = ctx.run_projection((k * 149) % nproj, N, mode=m, seed=4000 + 100 * j + k)
                    e[m].append(img.reshape(4, -1).sum(axis=1).astype(np.float64) / d)
            a = np.array(e[MODES[0]])
            for m in MODES[1:]:
                b = np.array(e[m])
                se = np.sqrt(a.var(axis=0, ddof=1) / K + b.var(axis=0, ddof=1) / K)
                # the runs of the two modes share their projection angles: compare angle by angle
                d = b - a
                zp = d.mean(axis=0) / (d.std(axis=0, ddof=1) / np.sqrt(K))
                print(f"  {m} / {MODES[0]} energy per history by class (primary, compton, rayleigh, multiple), {K} x {N:.0e} histories, {time.time() - t0:.0f} s:")
                print("    ratio", " ".join(f"{x:.6f}" for x in b.mean(axis=0) / a.mean(axis=0)), " paired z", " ".join(f"{x:+.2f}" for x in zp),
                      " relative sigma", " ".join(f"{x:.1e}" for x in (d.std(axis=0, ddof=1) / np.sqrt(K)) / a.mean(axis=0)))
