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


@pytest.mark.gpu
@pytest.mark.parametrize("hann,hann_y,off_x,wpc", [(0.0, 0.0, -150.0, None), (1.0, 1.0, -80.0, None), (0.7, 0.5, 0.0, (0.0, 1.05, 0.01)),
                                                 (1.0, 1.0, 150.0, None)])
def test_hip_fdk_matches_the_oracle(hann, hann_y, off_x, wpc):
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
    assert abs(vol[inside].mean() / mu - 1.0) < 0.01 and abs(vol[outside].mean()) < 0.03 * mu
