"""End-to-end demonstration (GPU): Monte Carlo scan of a small phantom -> air-normalised projection stack -> FDK volume,
i.e. the reference's `run-mc --reconstruct-3d` flow (scripts/run_mc_simulations.py:558-611) on the in-process engine.
Reduced detector (462 x 192 pixels of 1.552 mm over the reference's 717 x 298 mm, half-fan crop 256 columns), 180 projections.
Reports the orientation of the reconstruction relative to the phantom (best of the 16 axis-aligned candidates about the
rotation axis) and the reconstructed attenuation of water / bone / air."""
import sys, time, itertools
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, cases
pkg, eng = cases.pkg, cases.pkg.engine
M, recon = pkg.materials, pkg.reconstruction

shape, vs = (48, 48, 32), (5.0, 5.0, 5.0)
x, y, z = np.meshgrid(*[(np.arange(n) + 0.5 - n / 2) * s for n, s in zip(shape, vs)], indexing="ij", sparse=True)
mats = np.full(shape, M.material_number("air"), np.uint8); dens = np.full(shape, 0.0012, np.float32)
body = (x ** 2 + y ** 2 <= 100.0 ** 2) & (np.abs(z) <= 65); mats[body] = M.material_number("h2o"); dens[body] = 1.0
bone = ((x - 45) ** 2 + (y - 20) ** 2 <= 18.0 ** 2) & (np.abs(z) <= 40); mats[bone] = M.material_number("bone_050"); dens[bone] = 1.6
hole = ((x + 30) ** 2 + (y + 50) ** 2 <= 14.0 ** 2) & (np.abs(z - 10) <= 30); mats[hole] = M.material_number("air"); dens[hole] = 0.0012
geo = pkg.geometry.MCGeometry(mats, dens, vs)
wd = Path("/tmp/mc_to_recon"); wd.mkdir(exist_ok=True)
n_proj, det = 180, dict(n_detector_pixels=(462, 192), detector_size=(717.024, 297.984))
sim = pkg.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=int(4e7), n_projections=n_proj,
                                  angle_between_projections=360.0 / n_proj, **det)
inp = sim.prepare_simulation(wd, compress_geometry=False, engine=eng, binary_sidecar=True)
air = pkg.simulation.MCSimulation(pkg.geometry.MCAirGeometry(), cases.material_files(), cases.spectrum_file(), n_histories=int(2e9), n_projections=1, **det)
air_inp = air.prepare_simulation(wd / "air", compress_geometry=False, engine=eng)
t0 = time.time()
with eng.create(air_inp, device=0) as ctx:
    ctx.run_scan(mode="fast", crop_nx=256, output_folder=wd / "air", pixel_spacing=(1.552, 1.552))
with eng.create(inp, device=0) as ctx:
    r = ctx.run_scan(mode="fast", crop_nx=256, output_folder=wd, air_stack=wd / "air" / "projections_total.mha", air_sigma=(3.0, 3.0),
                     pixel_spacing=(1.552, 1.552))
print(f"scan: {time.time() - t0:.1f} s ({r['seconds_kernels']:.2f} s of kernels)")
rtk_geo = recon.create_geometry(n_proj, start_angle=90.0)   # cbctmc/mc/simulation.py:442-443
rtk_geo.write(wd / "geometry.xml")
dim, sp = (64, 48, 64), (4.0, 4.0, 4.0)
out, rep = recon.reconstruct_3d(wd / "projections_total_normalized.mha", wd / "geometry.xml", dimension=dim, spacing=sp, hann=1.0, hann_y=1.0)
vol, _, _ = recon.read_mha(out)          # [z_iec, y_iec, x_iec]
print(f"FDK: filter {rep['ms_filter']:.2f} ms, back-projection {rep['ms_backproject']:.2f} ms")
# phantom attenuation map on the reconstruction grid for each candidate orientation: IEC y = +-MC z; (x_iec, z_iec) = rotation/reflection of MC (x, y)
X, Y, Z = [-(n - 1) / 2 * s + s * np.arange(n) for n, s in zip(dim, sp)]
zi, yi, xi = np.meshgrid(Z, Y, X, indexing="ij")
def phantom_at(px, py, pz):
    ix = np.floor(px / vs[0] + shape[0] / 2).astype(int); iy = np.floor(py / vs[1] + shape[1] / 2).astype(int); iz = np.floor(pz / vs[2] + shape[2] / 2).astype(int)
    ok = (ix >= 0) & (ix < shape[0]) & (iy >= 0) & (iy < shape[1]) & (iz >= 0) & (iz < shape[2])
    d = np.where(ok, dens[np.clip(ix, 0, shape[0] - 1), np.clip(iy, 0, shape[1] - 1), np.clip(iz, 0, shape[2] - 1)], 0.0)
    return d
best = None
for sy in (1, -1):
    for (a, b, c, d) in ((1, 0, 0, 1), (0, 1, -1, 0), (-1, 0, 0, -1), (0, -1, 1, 0), (1, 0, 0, -1), (0, 1, 1, 0), (-1, 0, 0, 1), (0, -1, -1, 0)):
        ref = phantom_at(a * xi + b * zi, c * xi + d * zi, sy * yi)
        cc = np.corrcoef(ref.ravel(), vol.ravel())[0, 1]
        if best is None or cc > best[0]:
            best = (cc, sy, (a, b, c, d), ref)
cc, sy, mat, ref = best
print(f"best orientation: MC x = {mat[0]}*x_iec + {mat[1]}*z_iec, MC y = {mat[2]}*x_iec + {mat[3]}*z_iec, MC z = {sy}*y_iec; correlation {cc:.3f}")
for name, lo, hi in (("water", 0.99, 1.01), ("bone", 1.5, 1.7), ("air inside the field", -1, 0.01)):
    m = (ref > lo) & (ref < hi) & (np.abs(yi) < 40) & (xi ** 2 + zi ** 2 < 110 ** 2)
    print(f"  {name:22s} reconstructed mu = {vol[m].mean():.5f} /mm  ({m.sum()} voxels)")
