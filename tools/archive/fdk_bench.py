"""FDK reconstruction at the reference's size (GPU): 894 projections of 1024 x 768 pixels (0.388 mm, half-fan offset)
-> 464 x 250 x 464 voxels of 1 mm, hann = hannY = 1 (cbctmc/reconstruction/reconstruction.py:22-69 defaults).
Analytic sphere projections; prints kernel times, voxel updates per second and the recovered attenuation."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, cases, fdk_oracle as fo
recon = cases.pkg.reconstruction
n, nu, nv, du = (int(sys.argv[1]) if len(sys.argv) > 1 else 894), 1024, 768, 0.388
geo = recon.create_geometry(n, start_angle=90.0)
u0, v0 = -(nu - 1) / 2 * du, -(nv - 1) / 2 * du
mu, radius, centre = 0.02, 80.0, (15.0, 10.0, -20.0)
t0 = time.time()
proj = fo.sphere_projections(mu, radius, centre, n, nu, nv, du, du, u0, v0, geo.source_to_isocenter, geo.source_to_detector,
                             np.array(geo.gantry_angles), np.array(geo.projection_offsets_x), np.array(geo.projection_offsets_y)).astype(np.float32)
print(f"analytic projections: {time.time() - t0:.1f} s", flush=True)
dim, sp = (464, 250, 464), (1.0, 1.0, 1.0)
updates = n * dim[0] * dim[1] * dim[2]
for pad in (0.0, 1.0):  # 1.0 = the reference's default truncation correction: rows three times as long through the ramp
    for rep in range(2):
        t0 = time.time()
        vol, r = recon.fdk(proj, geo, (du, du), (u0, v0), dim, sp, hann=1.0, hann_y=1.0, pad=pad)
        wall = time.time() - t0
    print(f"pad {pad}: filter {r['ms_filter']:.1f} ms, backprojection {r['ms_backproject']:.1f} ms ({updates / r['ms_backproject'] / 1e6:.1f} G voxel updates/s), "
          f"wall incl. PCIe both ways {wall:.2f} s")
X, Y, Z = [-(k - 1) / 2 * s + s * np.arange(k) for k, s in zip(dim, sp)]
zz, yy, xx = np.meshgrid(Z, Y, X, indexing="ij")
rr = np.sqrt((xx - centre[0]) ** 2 + (yy - centre[1]) ** 2 + (zz - centre[2]) ** 2)
print(f"mu inside {vol[rr < radius - 10].mean():.6f} (true {mu}), outside {vol[(rr > radius + 10) & (rr < radius + 40)].mean():.6f}")
