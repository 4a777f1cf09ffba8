"""Row f4: FDK reconstruction.  CPU: the numpy oracle against analytic phantoms, RTK-style geometry XML, MetaImage I/O.
GPU: the HIP kernels (through the C ABI) against the oracle and the reference-shaped `reconstruct_3d` file flow.
Parity against RTK itself is unpinned (oracle/fdk_oracle.py header)."""
import sys
from pathlib import Path

import numpy as np
import pytest

import cases

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
import fdk_oracle as fo  # noqa: E402

recon = cases.pkg.reconstruction


def _half_fan_case(n=120, nu=96, nv=64, du=4.0, off_x=-80.0):
    geo = recon.create_geometry(n, start_angle=90.0, detector_offset_x=off_x)
    u0, v0 = -(nu - 1) / 2 * du, -(nv - 1) / 2 * du
    mu, radius, centre = 0.02, 60.0, (20.0, 5.0, -10.0)
    proj = fo.sphere_projections(mu, radius, centre, n, nu, nv, du, du, u0, v0, geo.source_to_isocenter, geo.source_to_detector,
                                 np.array(geo.gantry_angles), np.array(geo.projection_offsets_x), np.array(geo.projection_offsets_y))
    return geo, proj, (du, du), (u0, v0), (mu, radius, centre)


def _sphere_masks(dim, sp, centre, radius, margin):
    X, Y, Z = [-(n - 1) / 2 * s + s * np.arange(n) for n, s in zip(dim, sp)]
    zz, yy, xx = np.meshgrid(Z, Y, X, indexing="ij")
    r = np.sqrt((xx - centre[0]) ** 2 + (yy - centre[1]) ** 2 + (zz - centre[2]) ** 2)
    return r < radius - margin, (r > radius + margin) & (r < radius + 50)


def test_oracle_recovers_a_uniform_sphere_half_fan_and_centred():
    """Scaling constants, angular weights, cosine and displaced-detector weights: mu inside within 0.5 %, ~0 outside."""
    dim, sp = (64, 40, 64), (4.0, 4.0, 4.0)
    for off_x in (-150.0, -80.0, 150.0, 0.0):  # -150: 40 mm of overlap on a 380 mm detector, the reference's half-fan proportions
        geo, proj, (du, dv), (u0, v0), (mu, radius, centre) = _half_fan_case(off_x=off_x)
        vol = fo.reconstruct(proj, du, dv, u0, v0, geo.source_to_isocenter, geo.source_to_detector, geo.gantry_angles, geo.projection_offsets_x,
                             geo.projection_offsets_y, dim, sp)
        inside, outside = _sphere_masks(dim, sp, centre, radius, 12.0)
        assert abs(vol[inside].mean() / mu - 1.0) < 5e-3, (off_x, vol[inside].mean())
        assert abs(vol[outside].mean()) < 0.03 * mu, (off_x, vol[outside].mean())


def test_symmetric_padding():
    """An off-centre detector is padded with zero columns until it is symmetric about the central ray: the filtered rows
    are needed beyond the physical edge (without the padding, voxels outside the overlap radius come out 35-70 % too high)."""
    assert fo.symmetric_padding(96, 4.0, -190.0, 0.0, 0.0) == (0, 0)
    assert fo.symmetric_padding(96, 4.0, -190.0, -150.0, -150.0) == (0, 75)   # [-340, 40] -> [-340, 340]
    assert fo.symmetric_padding(96, 4.0, -190.0, 150.0, 150.0) == (75, 0)
    assert fo.symmetric_padding(1024, 0.388, -198.462, -159.856, -159.856) == (0, 824)  # reference half-fan: 1848 columns


def test_displaced_weights_are_complementary():
    u = np.linspace(-270.0, 110.0, 96)
    w = fo.displaced_weights(u, 1500.0)
    assert np.all(w[u < -110.0] == 1.0) and np.all((w >= 0) & (w <= 1))
    inner = np.abs(u) <= 110.0
    wi = np.interp(-u[inner], u, w)  # weight of the conjugate ray
    assert np.allclose(w[inner] + wi, 1.0, atol=2e-3)
    assert np.all(fo.displaced_weights(np.linspace(-100, 100, 51), 1500.0) == 0.5)


def test_hann_windows():
    h = fo.ramp_kernel(64, 0.0)
    assert h[64] == 0.25 and h[65] == pytest.approx(-1 / np.pi ** 2) and h[66] == 0.0
    assert abs(h.sum()) < 2e-3  # the ramp has no DC response (up to truncation)
    hh = fo.ramp_kernel(64, 1.0)
    assert np.allclose(hh[1:-1], 0.25 * h[:-2] + 0.5 * h[1:-1] + 0.25 * h[2:], atol=2e-5)  # Hann at Nyquist = [1/4, 1/2, 1/4] smoothing
    assert np.allclose(fo.hann_y_kernel(1.0), [0.25, 0.5, 0.25])
    assert abs(fo.hann_y_kernel(0.5).sum() - 1.0) < 1e-3


def test_geometry_xml_round_trip_and_matrix(tmp_path):
    geo = recon.create_geometry(7, start_angle=90.0)
    assert geo.source_to_isocenter == 1000.0 and geo.source_to_detector == 1500.0 and geo.projection_offsets_x[0] == -159.856
    back = recon.CircularGeometry.read(geo.write(tmp_path / "geometry.xml"))
    assert back == geo
    text = (tmp_path / "geometry.xml").read_text()
    assert text.startswith('<?xml version="1.0"?>\n<!DOCTYPE RTKGEOMETRY>\n<RTKThreeDCircularGeometry version="3">')
    # the matrix maps a point to the stack coordinates used by the oracle / the kernels
    p = np.array([30.0, -12.0, 45.0, 1.0])
    for i in (0, 3):
        t = np.deg2rad(geo.gantry_angles[i])
        xr, zr = p[0] * np.cos(t) - p[2] * np.sin(t), p[0] * np.sin(t) + p[2] * np.cos(t)
        mag = geo.source_to_detector / (geo.source_to_isocenter - zr)
        uvw = geo.matrix(i) @ p
        assert np.allclose(uvw[:2] / uvw[2], [mag * xr - geo.projection_offsets_x[i], mag * p[1] - geo.projection_offsets_y[i]])


def test_metaimage_round_trip(tmp_path):
    v = np.random.default_rng(0).normal(size=(5, 4, 3)).astype(np.float32)
    recon.write_mha(tmp_path / "v.mha", v, (1.0, 2.0, 3.0), (-1.0, -3.0, -6.0))
    a, sp, org = recon.read_mha(tmp_path / "v.mha")
    assert np.array_equal(a, v) and sp == [1.0, 2.0, 3.0] and org == [-1.0, -3.0, -6.0]


def _truncated_sphere_case(n=120, nu=96, nv=32, du=4.0):
    """A sphere wider than the field of view of a centred detector (radius 170 mm, FOV radius 128 mm): every row is truncated."""
    geo = recon.create_geometry(n, start_angle=90.0, detector_offset_x=0.0)
    u0, v0 = -(nu - 1) / 2 * du, -(nv - 1) / 2 * du
    mu, radius = 0.02, 170.0
    proj = fo.sphere_projections(mu, radius, (0.0, 0.0, 0.0), n, nu, nv, du, du, u0, v0, geo.source_to_isocenter, geo.source_to_detector,
                                 np.array(geo.gantry_angles), np.array(geo.projection_offsets_x), np.array(geo.projection_offsets_y))
    assert proj[0, nv // 2, 0] > 0.5 * proj[0, nv // 2, nu // 2]  # the rows end far above zero
    return geo, proj, (du, du), (u0, v0), mu


def _ring_means(vol, dim, sp, rings):
    X = -(dim[0] - 1) / 2 * sp[0] + sp[0] * np.arange(dim[0])
    zz, xx = np.meshgrid(X, X, indexing="ij")
    r = np.sqrt(xx ** 2 + zz ** 2)
    mid = vol[:, dim[1] // 2, :]
    return [float(mid[(r >= lo) & (r < hi)].mean()) for lo, hi in rings]


def test_truncation_extension_rule():
    """rtkfdk --pad: point reflection about the border value, feathered by sin^0.75 (oracle header: restated from RTK's
    published filter, parity unpinned)."""
    rows, nxt = fo.truncation_extension(np.arange(1.0, 9.0)[None], 0.5)
    assert nxt == 4 and rows.shape == (1, 16)
    w = np.sin((4 - np.arange(1, 5)) * np.pi / 6.0) ** 0.75
    assert np.allclose(rows[0, 4:12], np.arange(1.0, 9.0))
    assert np.allclose(rows[0, 3::-1], w * (2 * 1.0 - np.arange(2.0, 6.0)))     # left of column 0: 2 p(0) - p(d)
    assert np.allclose(rows[0, 12:], w * (2 * 8.0 - np.arange(7.0, 3.0, -1.0)))  # right of the last column
    assert rows[0, 0] == 0 and rows[0, -1] == 0                                  # feathered to zero at the far ends
    same, none = fo.truncation_extension(np.ones((3, 5)), 0.0)
    assert none == 0 and same.shape == (3, 5)
    _, capped = fo.truncation_extension(np.ones((2, 6)), 4.0)
    assert capped == 5                                                           # never more than the row can mirror


def test_oracle_truncation_correction_flattens_a_truncated_object():
    """Without --pad the ramp rings at the cut-off edge of every row: the reconstruction rises towards the edge of the field
    of view (here +26 % from the centre to the outer ring).  With the reference's --pad the profile is flat within a few
    percent (the remaining uniform offset is the part of the object that was never measured)."""
    geo, proj, (du, dv), (u0, v0), mu = _truncated_sphere_case()
    dim, sp = (48, 8, 48), (5.0, 5.0, 5.0)
    rings = ((0, 40), (80, 120))
    flat = {}
    for pad in (0.0, 0.5, 1.0):
        vol = fo.reconstruct(proj, du, dv, u0, v0, geo.source_to_isocenter, geo.source_to_detector, geo.gantry_angles, geo.projection_offsets_x,
                             geo.projection_offsets_y, dim, sp, hann=1.0, pad=pad)
        centre, outer = _ring_means(vol, dim, sp, rings)
        flat[pad] = outer / centre
        assert 0.75 * mu < centre < 1.2 * mu
    assert flat[0.0] > 1.2
    assert abs(flat[0.5] - 1.0) < 0.08 and abs(flat[1.0] - 1.0) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("pad,off_x,direct", [(1.0, 0.0, False), (0.5, -150.0, False), (0.3, 80.0, True)])
def test_hip_fdk_truncation_correction_matches_the_oracle(pad, off_x, direct, monkeypatch):
    """extend_rows_kernel + the longer ramp against the oracle's truncation_extension, centred and half-fan, both ramp routes."""
    if direct:
        monkeypatch.setenv("MCGPU_FDK_DIRECT_RAMP", "1")
    if off_x == 0.0:
        geo, proj, (du, dv), (u0, v0), mu = _truncated_sphere_case(n=90)
    else:
        geo, proj, (du, dv), (u0, v0), _ = _half_fan_case(n=90, off_x=off_x)
    rng = np.random.default_rng(7)
    proj = proj + 0.05 * rng.normal(size=proj.shape)
    dim, sp = (48, 12, 40), (5.0, 5.0, 6.0)
    want = fo.reconstruct(proj.astype(np.float32), du, dv, u0, v0, geo.source_to_isocenter, geo.source_to_detector, geo.gantry_angles,
                          geo.projection_offsets_x, geo.projection_offsets_y, dim, sp, hann=1.0, hann_y=1.0, pad=pad)
    got, _ = recon.fdk(proj, geo, (du, dv), (u0, v0), dim, sp, hann=1.0, hann_y=1.0, pad=pad)
    plain, _ = recon.fdk(proj, geo, (du, dv), (u0, v0), dim, sp, hann=1.0, hann_y=1.0, pad=0.0)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 3e-4 * scale, np.abs(got - want).max() / scale
    if off_x == 0.0:  # truncated rows: the correction really changes the answer (the half-fan sphere lies inside its field of view)
        assert np.abs(got - plain).max() > 1e-2 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("hann,hann_y,off_x,wpc", [(0.0, 0.0, -150.0, None), (1.0, 1.0, -80.0, None), (0.7, 0.5, 0.0, (0.0, 1.05, 0.01)),
                                                 (1.0, 1.0, 150.0, None)])
def test_hip_fdk_matches_the_oracle(hann, hann_y, off_x, wpc, monkeypatch):
    if off_x == 150.0:
        monkeypatch.setenv("MCGPU_FDK_DIRECT_RAMP", "1")  # the direct LDS convolution instead of the hipFFT ramp
    geo, proj, (du, dv), (u0, v0), (mu, radius, centre) = _half_fan_case(n=90, off_x=off_x)
    rng = np.random.default_rng(5)
    proj = proj + 0.05 * rng.normal(size=proj.shape)  # noise: every filter tap and interpolation weight matters
    dim, sp = (48, 30, 40), (5.0, 5.0, 6.0)
    want = fo.reconstruct(proj.astype(np.float32), du, dv, u0, v0, geo.source_to_isocenter, geo.source_to_detector, geo.gantry_angles,
                          geo.projection_offsets_x, geo.projection_offsets_y, dim, sp, hann=hann, hann_y=hann_y, wpc=wpc)
    got, rep = recon.fdk(proj, geo, (du, dv), (u0, v0), dim, sp, hann=hann, hann_y=hann_y, water_pre_correction=wpc)
    scale = np.abs(want).max()
    assert got.shape == want.shape and scale > 0
    assert np.abs(got - want).max() < 2e-4 * scale, np.abs(got - want).max() / scale  # float32 kernels vs float64 oracle
    assert rep["ms_backproject"] > 0


def test_angular_gaps_rule():
    assert np.allclose(fo.angular_gaps(np.arange(0, 360, 4.0)), np.deg2rad(4.0))
    g = fo.angular_gaps([0.0, 10.0, 30.0, 200.0, 350.0])
    assert np.allclose(np.rad2deg(g), [10.0, 15.0, 95.0, 160.0, 80.0]) and abs(g.sum() - 2 * np.pi) < 1e-12
    assert np.allclose(np.rad2deg(fo.angular_gaps([5.0, 5.0, 185.0])), [90.0, 90.0, 180.0])  # duplicates share


@pytest.mark.gpu
def test_hip_fdk_weights_projections_by_their_angular_gaps():
    """A non-uniform angle set (a 60-degree wedge of views removed, two views doubled): the kernels weight every projection
    by its angular gap like the oracle (and like rtkfdk, which takes the gaps from the geometry file), not by 2 pi / n."""
    geo, proj, (du, dv), (u0, v0), _ = _half_fan_case(n=120, off_x=-80.0)
    keep = np.array([k for k in range(120) if not (40 <= k < 60)] + [3, 77])
    geo.gantry_angles = list(np.asarray(geo.gantry_angles)[keep])
    geo.projection_offsets_x = list(np.asarray(geo.projection_offsets_x)[keep])
    geo.projection_offsets_y = list(np.asarray(geo.projection_offsets_y)[keep])
    proj = proj[keep]
    dim, sp = (40, 24, 32), (6.0, 6.0, 7.0)
    want = fo.reconstruct(proj.astype(np.float32), du, dv, u0, v0, geo.source_to_isocenter, geo.source_to_detector, geo.gantry_angles,
                          geo.projection_offsets_x, geo.projection_offsets_y, dim, sp, hann=1.0, hann_y=0.0)
    got, _ = recon.fdk(proj, geo, (du, dv), (u0, v0), dim, sp, hann=1.0, hann_y=0.0)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 2e-4 * scale, np.abs(got - want).max() / scale


@pytest.mark.gpu
def test_reconstruct_3d_file_flow(tmp_path):
    """Reference-shaped call: normalised stack + geometry.xml in, recon_fdk3d.mha (+ .yaml) out; sphere value recovered."""
    geo, proj, (du, dv), (u0, v0), (mu, radius, centre) = _half_fan_case(n=180, off_x=-150.0)
    recon.write_mha(tmp_path / "projections_total_normalized.mha", proj.astype(np.float32), (du, dv, 1.0), (u0, v0, 0.0))
    geo.write(tmp_path / "geometry.xml")
    dim, sp = (64, 40, 64), (4.0, 4.0, 4.0)
    out, rep = recon.reconstruct_3d(tmp_path / "projections_total_normalized.mha", tmp_path / "geometry.xml", dimension=dim, spacing=sp,
                                    hann=1.0, hann_y=1.0)
    assert out == tmp_path / "reconstructions" / "recon_fdk3d.mha" and out.with_suffix(".yaml").exists()
    vol, vsp, vorg = recon.read_mha(out)
    assert vol.shape == (64, 40, 64) and vsp == [4.0, 4.0, 4.0] and vorg == [-126.0, -78.0, -126.0]
    inside, outside = _sphere_masks(dim, sp, centre, radius, 16.0)
    # the reference's default pad = 1.0 is in force: the rows of this sphere end at zero, so nothing is extended, but the
    # Hann-apodised ramp is cut off three times further out than without padding (1.2 % instead of 0.9 % here)
    assert abs(vol[inside].mean() / mu - 1.0) < 0.015 and abs(vol[outside].mean()) < 0.03 * mu


@pytest.mark.gpu
def test_mc_scan_to_fdk_round_trip(engine, tmp_path):
    """The reference's `run-mc --reconstruct-3d` flow end to end on the engine: Monte Carlo scan of a water cylinder with a
    bone rod and an air hole -> air-normalised stack (half-fan crop) -> RTK-style geometry (start angle 90 degrees,
    cbctmc/mc/simulation.py:442-443) -> FDK.  The volume must show the phantom in the orientation MC (x, y, z) =
    (x, -z, -y) of RTK's IEC frame -- i.e. the Monte Carlo geometry, the stack conventions (z flip, crop of the first columns)
    and the reconstruction geometry fit together -- with plausible attenuation values (scatter and beam hardening included)."""
    pkg, M = cases.pkg, cases.pkg.materials
    shape, vs = (48, 48, 32), (5.0, 5.0, 5.0)
    x, y, z = np.meshgrid(*[(np.arange(n) + 0.5 - n / 2) * s for n, s in zip(shape, vs)], indexing="ij", sparse=True)
    mats = np.full(shape, M.material_number("air"), np.uint8)
    dens = np.full(shape, 0.0012, np.float32)
    body = (x ** 2 + y ** 2 <= 100.0 ** 2) & (np.abs(z) <= 65)
    mats[body], dens[body] = M.material_number("h2o"), 1.0
    bone = ((x - 45) ** 2 + (y - 20) ** 2 <= 18.0 ** 2) & (np.abs(z) <= 40)
    mats[bone], dens[bone] = M.material_number("bone_050"), 1.6
    hole = ((x + 30) ** 2 + (y + 50) ** 2 <= 14.0 ** 2) & (np.abs(z - 10) <= 30)
    mats[hole], dens[hole] = M.material_number("air"), 0.0012
    n_proj, det = 120, dict(n_detector_pixels=(462, 192), detector_size=(717.024, 297.984))
    sim = pkg.simulation.MCSimulation(pkg.geometry.MCGeometry(mats, dens, vs), cases.material_files(), cases.spectrum_file(), n_histories=int(1.5e7),
                                      n_projections=n_proj, angle_between_projections=360.0 / n_proj, **det)
    inp = sim.prepare_simulation(tmp_path, compress_geometry=False, engine=engine, binary_sidecar=True)
    air = pkg.simulation.MCSimulation(pkg.geometry.MCAirGeometry(), cases.material_files(), cases.spectrum_file(), n_histories=int(1e9), n_projections=1, **det)
    air_inp = air.prepare_simulation(tmp_path / "air", compress_geometry=False, engine=engine)
    with engine.create(air_inp, device=0) as ctx:
        ctx.run_scan(mode="fast", crop_nx=256, output_folder=tmp_path / "air", pixel_spacing=(1.552, 1.552))
    with engine.create(inp, device=0) as ctx:
        ctx.run_scan(mode="fast", crop_nx=256, output_folder=tmp_path, air_stack=tmp_path / "air" / "projections_total.mha", air_sigma=(3.0, 3.0),
                     pixel_spacing=(1.552, 1.552))
    recon.create_geometry(n_proj, start_angle=90.0).write(tmp_path / "geometry.xml")
    dim, sp = (64, 48, 64), (4.0, 4.0, 4.0)
    out, _ = recon.reconstruct_3d(tmp_path / "projections_total_normalized.mha", tmp_path / "geometry.xml", dimension=dim, spacing=sp)
    vol, _, _ = recon.read_mha(out)
    X, Y, Z = [-(n - 1) / 2 * s + s * np.arange(n) for n, s in zip(dim, sp)]
    zi, yi, xi = np.meshgrid(Z, Y, X, indexing="ij")

    def phantom_on_grid(px, py, pz):  # MC coordinates of the reconstruction grid -> phantom density there
        ix, iy, iz = [np.floor(p / s + n / 2).astype(int) for p, s, n in zip((px, py, pz), vs, shape)]
        ok = (ix >= 0) & (ix < shape[0]) & (iy >= 0) & (iy < shape[1]) & (iz >= 0) & (iz < shape[2])
        return np.where(ok, dens[np.clip(ix, 0, shape[0] - 1), np.clip(iy, 0, shape[1] - 1), np.clip(iz, 0, shape[2] - 1)], 0.0)

    ref = phantom_on_grid(xi, -zi, -yi)
    cc = np.corrcoef(ref.ravel(), vol.ravel())[0, 1]
    assert cc > 0.85, cc  # noisy scan (1.5e7 histories per projection); tools/archive/mc_to_recon.py reaches 0.96 with more
    central = (np.abs(yi) < 40) & (xi ** 2 + zi ** 2 < 110 ** 2)
    water, rod, air_hole = [vol[central & (ref > lo) & (ref < hi)].mean() for lo, hi in ((0.99, 1.01), (1.5, 1.7), (-1, 0.01))]
    assert 0.015 < water < 0.022 and rod > 1.4 * water and air_hole < 0.25 * water, (water, rod, air_hole)
    for wrong in (phantom_on_grid(xi, zi, -yi), phantom_on_grid(-xi, -zi, -yi)):  # mirrored in-plane: the rod is not where these expect it
        assert vol[central & (wrong > 1.5) & (wrong < 1.7)].mean() < 1.2 * water


def test_ramp_frequency_response_with_and_without_the_hann_cut():
    """What RTK documents for `rtkfdk --hann` (FFTRampImageFilter: the ramp |f| multiplied by a Hann window that reaches zero at
    hann x Nyquist): no DC gain with or without the window, the response follows |f| 0.5 (1 + cos(pi f / fc)) below the cut and
    vanishes above it.  Holds the oracle's kernel (the one csrc/fdk.hip is compared with) to the closed form; parity with RTK's own
    numbers stays unpinned (no RTK here)."""
    for hann in (0.0, 1.0, 0.7, 0.4):
        n_half = 256
        h = fo.ramp_kernel(n_half, hann)
        assert abs(h.sum()) < 1e-3, hann                                       # DC gain: a constant row filters to zero
        m = 8192
        buf = np.zeros(m)
        buf[: n_half + 1] = h[n_half:]
        buf[-n_half:] = h[:n_half]
        H = np.real(np.fft.fft(buf))
        f = np.abs(np.fft.fftfreq(m))
        fc = 0.5 * hann if hann > 0 else 0.5
        want = f * (0.5 * (1.0 + np.cos(np.pi * f / fc)) if hann > 0 else 1.0)
        want = np.where(f < fc, want, 0.0) if hann > 0 else f
        assert np.max(np.abs(H - want)) < 2.5e-3, (hann, float(np.max(np.abs(H - want))))   # truncation of the kernel to 513 taps
        low = (f > 0.01) & (f < 0.05)
        assert np.allclose(H[low] / f[low], (0.5 * (1.0 + np.cos(np.pi * f[low] / fc)) if hann > 0 else 1.0), atol=0.03)


def test_geometry_matrix_closed_form_and_the_reference_angle_series(tmp_path):
    """`create_geometry` (cbctmc/forward_projection.py:152-199) calls AddProjection(sid, sdd, start + i arc / n, offset_x, offset_y) for
    every projection; RTK's documented matrix of such a projection (no source offsets, no in-plane / out-of-plane angles) is
    T(-offset) . [[-sdd, 0, 0, 0], [0, -sdd, 0, 0], [0, 0, 1, -sid]] . R_y(-angle), in closed form
        [ -sdd cos - ox sin,    0,  sdd sin - ox cos,  ox sid ]
        [        - oy sin,   -sdd,         - oy cos,   oy sid ]
        [             sin,      0,              cos,    - sid ]
    The XML this package writes carries that matrix, digit for digit what it computes, for the reference's default half-fan scan."""
    n, start, arc = 894, 270.0, 360.0
    geo = recon.create_geometry(n, start_angle=start)
    d = cases_defaults()
    sid, sdd, ox, oy = d.source_to_isocenter_distance, d.source_to_detector_distance, d.detector_lateral_displacement, 0.0
    assert (geo.source_to_isocenter, geo.source_to_detector) == (sid, sdd)
    angles = np.array([(start + i * arc / n) % 360.0 for i in range(n)])
    assert np.allclose(geo.gantry_angles, angles, rtol=0, atol=1e-12) and geo.projection_offsets_x == [ox] * n and geo.projection_offsets_y == [oy] * n
    text = geo.write(tmp_path / "geometry.xml").read_text()
    blocks = text.split("<Matrix>")[1:]
    assert len(blocks) == n
    for i in (0, 1, 223, 447, 893):
        t = np.deg2rad(angles[i])
        c, s = np.cos(t), np.sin(t)
        closed = np.array([[-sdd * c - ox * s, 0.0, sdd * s - ox * c, ox * sid], [-oy * s, -sdd, -oy * c, oy * sid], [s, 0.0, c, -sid]])
        assert np.allclose(geo.matrix(i), closed, rtol=0, atol=1e-9), i
        rows = np.array([[float(v) for v in line.split()] for line in blocks[i].split("</Matrix>")[0].strip().split("\n")])
        assert rows.shape == (3, 4) and np.allclose(rows, closed, rtol=1e-13, atol=1e-9), i
        # the source sits at distance sid on the rotated z axis: the matrix sends it to the point at infinity (third component 0)
        assert abs((closed @ np.array([sid * s, 0.0, sid * c, 1.0]))[2]) < 1e-9
    # isocentre -> (-offset_x, -offset_y) in detector coordinates, whatever the angle
    for i in (0, 300, 893):
        uvw = geo.matrix(i) @ np.array([0.0, 0.0, 0.0, 1.0])
        assert np.allclose(uvw[:2] / uvw[2], [-ox, -oy])


def cases_defaults():
    import cases
    return cases.pkg.defaults.DEFAULTS


def test_rocm_container_recipe_matches_the_build():
    """docker/Dockerfile.rocm has never produced an image (no docker here): a static check that what it COPYs exists, that what it
    installs is what `make -C 4d-cbct-mc_amd/csrc` builds, that the make variable it sets exists, and that the `mpirun` shim the
    reference's command line needs (cbctmc/mc/simulation.py:187-198) is part of it."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    text = (root / "docker" / "Dockerfile.rocm").read_text()
    lines = [l.strip() for l in text.replace("\\\n", " ").split("\n") if l.strip() and not l.strip().startswith("#")]
    assert lines[0].startswith("FROM rocm/") and any(l.startswith("WORKDIR ") for l in lines)
    copies = [l.split()[1:] for l in lines if l.startswith("COPY ")]
    assert copies
    for src, dst in copies:
        assert (root / src).exists(), f"COPY source {src} is not in the repository"
    makefile = (root / "4d-cbct-mc_amd" / "csrc" / "Makefile").read_text()
    run = " ".join(l for l in lines if l.startswith("RUN "))
    m = re.search(r"make -C (\S+) .*?ARCH=(\w+)", run)
    assert m and (root / m.group(1) / "Makefile").is_file() and m.group(2) == "gfx950" and re.search(r"^ARCH\s*\?=", makefile, re.M)
    assert any(src == m.group(1) for src, _ in copies) and any(src == "include" for src, _ in copies)   # the Makefile reads ../../include
    built = {"4d-cbct-mc_amd/" + re.search(rf"^{var} := \.\./(\S+)", makefile, re.M).group(1) for var in ("LIB", "EXE")}
    installed = set(re.findall(r"install -m 0755 (\S+)", run))
    assert installed == built, (installed, built)
    assert "/usr/local/bin/MC-GPU_v1.3.x" in run and "ldconfig" in run
    assert ["docker/mpirun", "/usr/local/bin/mpirun"] in copies and "chmod 0755 /usr/local/bin/mpirun" in run
    assert any(l.startswith("ENV ") and "HSA_ENABLE_IPC_MODE_LEGACY=0" in l for l in lines)
    # everything the csrc Makefile compiles lives in the directories the image copies
    for src in re.findall(r"^\S+\.o: (\S+)", makefile, re.M):
        assert (root / "4d-cbct-mc_amd" / "csrc" / src).is_file(), src
