#!/usr/bin/env python3
"""Dev measurement (GPU): FAST kernel throughput on a body-filling, thorax-like 512x512x256 volume (BASELINE config 4 shape;
synthetic ellipsoids, SURVEY.md 8d input 3) -- the case where the exterior hop helps least."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
pkg, eng = cases.pkg, cases.pkg.engine
M = pkg.materials
shape = (512, 512, 256)
x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float32) - n / 2 for n in shape], indexing="ij", sparse=True)
mats = np.full(shape, M.material_number("air"), np.uint8); dens = np.full(shape, 0.0013, np.float32)
def ell(cx, cy, cz, ax, ay, az): return ((x - cx) / ax) ** 2 + ((y - cy) / ay) ** 2 + ((z - cz) / az) ** 2 <= 1.0
body = ell(0, 0, 0, 175, 125, 400); mats[body] = M.material_number("h2o"); dens[body] = 1.0
fat = body & ~ell(0, 0, 0, 160, 110, 400); mats[fat] = M.material_number("ldpe"); dens[fat] = 0.92
for sx in (-1, 1):
    lung = ell(sx * 75, -10, 0, 60, 80, 110); mats[lung] = M.material_number("h2o"); dens[lung] = 0.26
spine = ell(0, 85, 0, 22, 22, 400); mats[spine] = M.material_number("bone_050"); dens[spine] = 1.4
for k in range(-5, 6):
    rib = (ell(0, 0, k * 22, 150, 105, 5) & ~ell(0, 0, k * 22, 140, 95, 5)); mats[rib] = M.material_number("bone_020"); dens[rib] = 1.14
geo = pkg.geometry.MCGeometry(mats, dens, (1.0, 1.0, 1.0))
wd = Path("/tmp/mcgpu_thorax"); wd.mkdir(exist_ok=True)
sim = pkg.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=int(1e8), n_projections=894, angle_between_projections=360.0 / 894)
t0 = time.time(); inp = sim.prepare_simulation(wd, compress_geometry=False, engine=eng, binary_sidecar=True); t1 = time.time()
with eng.create(inp, device=0) as ctx:
    t2 = time.time()
    out = {"prepare_s": round(t1 - t0, 1), "load_s": round(t2 - t1, 2), "bricks": ctx.geti("brick_count"), "mixed": ctx.geti("bricks_mixed"), "exterior": ctx.geti("bricks_exterior")}
    for p in (0, 223, 447):
        ctx.run_projection(p, int(2e7), mode="fast", seed=42)
        _, secs, done = ctx.run_projection(p, int(1e8), mode="fast", seed=42)
        out[f"p{p}_hist_per_s"] = round(done / secs / 1e9, 3)
    _, secs, done = ctx.run_projection(0, int(3e7), mode="stats", seed=42)
    s = ctx.scheduler_stats()
    out["iter_per_hist"] = round(s["iterations"] / done, 3); out["flying"] = round(s["flying_lanes"] / max(s["iterations"], 1), 1)
    out["cycles"] = {k[7:]: round(s[k] / done) for k in s if k.startswith("cycles_")}
print(json.dumps(out))
