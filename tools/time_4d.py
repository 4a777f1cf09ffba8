import sys, time, numpy as np
from pathlib import Path
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cases
pkg, eng = cases.pkg, cases.pkg.engine
M = pkg.materials
shape = (305, 300, 152)
x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float32) - n / 2 for n in shape], indexing="ij", sparse=True)
mats = np.full(shape, M.material_number("air"), np.uint8); dens = np.full(shape, 0.0013, np.float32)
body = ((x / 140) ** 2 + (y / 100) ** 2 <= 1) & (np.abs(z) < 70); mats[body] = M.material_number("h2o"); dens[body] = 1.0
lung = (((x - 60) / 45) ** 2 + (y / 60) ** 2 + (z / 55) ** 2 <= 1) | (((x + 60) / 45) ** 2 + (y / 60) ** 2 + (z / 55) ** 2 <= 1); dens[lung] = 0.26
sp = (x ** 2 + (y - 70) ** 2 <= 15 ** 2) & (np.abs(z) < 70); mats[sp] = M.material_number("bone_050"); dens[sp] = 1.4
geo = pkg.geometry.MCGeometry(mats, dens, (1.0, 1.0, 1.0))
wd = Path("/tmp/mcgpu_cirs"); wd.mkdir(exist_ok=True)
sim = pkg.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=int(1e8), projection_angles=[270.0, 270.4], angle_between_projections=0.4)
inp = sim.prepare_simulation(wd, compress_geometry=False, engine=eng, binary_sidecar=True)
u = np.zeros((3,) + shape[::-1], np.float32); u[2] = 3.3
with eng.create(inp, device=0) as ctx:
    mz, dz = np.ascontiguousarray(np.transpose(mats, (2, 1, 0))), np.ascontiguousarray(np.transpose(dens, (2, 1, 0)))
    for rep in range(3):
        t0 = time.time(); m2, d2 = ctx.warp_volume(mz, dz, u, 1, 0.0013); t1 = time.time()
        g2 = pkg.geometry.MCGeometry(np.transpose(m2, (2, 1, 0)), np.transpose(d2, (2, 1, 0)), (1.0, 1.0, 1.0)); t2 = time.time()
        ctx.set_geometry(g2); t3 = time.time()
        ctx.set_projection_angles([270.0, 270.0, 280.0, 290.0]); t4 = time.time()
        print(f"warp {t1-t0:.3f} s, numpy transposes {t2-t1:.3f} s, set_geometry {t3-t2:.3f} s, set_angles {t4-t3:.4f} s")
    _, secs, done = ctx.run_projection(1, int(1e8), mode="fast", seed=42)
    _, secs, done = ctx.run_projection(2, int(1e8), mode="fast", seed=42)
    print("CIRS-like FAST rate", done / secs / 1e9, "e9 hist/s")
