#!/usr/bin/env python3
"""Dev measurement (GPU): cost of one respiratory-state change on the bundled CIRS phantom (305 x 300 x 152 voxels):
on the device (mcgpu_warp_geometry) against the route through the host (warp_volume + set_geometry)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
pkg, eng = cases.pkg, cases.pkg.engine
geo = pkg.geometry.MCCIRSPhantomGeometry.from_base_geometry().place_insert()
shape = geo.image_shape
wd = Path("/tmp/mcgpu_cirs4d"); wd.mkdir(exist_ok=True)
sim = pkg.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=int(1e8), projection_angles=[270.0, 270.4], angle_between_projections=0.4)
inp = sim.prepare_simulation(wd, compress_geometry=False, engine=eng, binary_sidecar=True)
z = np.linspace(-1, 1, shape[2], dtype=np.float32)[None, None, :]
with eng.create(inp, device=0) as ctx:
    for rep in range(4):
        field = np.zeros((3,) + shape, np.float32)
        field[2] = (3.0 + rep) * (1 - z * z)  # SI motion, up to 15 mm in the reference's use (SURVEY 8d input 4)
        t0 = time.perf_counter(); ctx.warp_geometry(field, frame="geometry"); t1 = time.perf_counter()
        ctx.set_projection_angles([270.0, 270.0, 280.0, 290.0]); t2 = time.perf_counter()
        print(f"device route: warp_geometry {1e3*(t1-t0):.1f} ms (field upload {field.nbytes/1e6:.0f} MB included), set_angles {1e3*(t2-t1):.2f} ms")
    model = type("M", (), {"predict": lambda self, x: field})()
    sim4 = pkg.simulation.MCSimulation4D(model, geo, cases.material_files(), cases.spectrum_file())
    for rep in range(2):
        t0 = time.perf_counter(); g2 = sim4.warp_geometry(ctx, 0.0, 0.0); t1 = time.perf_counter(); ctx.set_geometry(g2); t2 = time.perf_counter()
        print(f"host route: warp_volume + numpy {1e3*(t1-t0):.1f} ms, set_geometry {1e3*(t2-t1):.1f} ms")
    _, secs, done = ctx.run_projection(1, int(1e8), mode="fast", seed=42)
    _, secs, done = ctx.run_projection(2, int(1e8), mode="fast", seed=42)
    print("CIRS FAST rate", round(done / secs / 1e9, 2), "e9 hist/s")
