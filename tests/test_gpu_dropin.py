"""GPU tests of the drop-in boundary: the `MC-GPU_v1.3.x <input.in>` executable and the Python mirror of
cbctmc.mc.MCSimulation, consumed the way cbctmc/mc/projection.py and simulation.py consume the reference."""
import re
import subprocess

import numpy as np
import pytest

import cases
import oracle_lib as ol
import parity

pytestmark = pytest.mark.gpu


def _read_like_reference(path, nz, nx, dtype=np.float32):
    """cbctmc/mc/projection.py:36-51 without the half-fan crop (the reference casts to float32)."""
    data = np.loadtxt(path, dtype=np.float64).astype(dtype)
    return np.flip(data.reshape(nz, nx, 4), axis=0)


def test_executable_compat_matches_oracle_and_log_contract(engine, tmp_path):
    inp = cases.build_case("catphan64_ct", tmp_path, n_histories=19200 * 3)
    exe = engine.EXE_PATH
    assert exe.exists()
    res = subprocess.run([str(exe), str(inp), "--mode", "compat"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert not re.search("(?i)error", res.stdout), "the caller greps the log for 'error' (sim.py:204)"
    found = re.findall(r"Simulating Projection (\d{1,4}) of (\d{1,4})", res.stdout)
    assert [int(a) for a, _ in found] == [1, 2, 3, 4] and {int(b) for _, b in found} == {4}
    files = sorted(f.name for f in tmp_path.iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert files == ["projection_270.000000deg", "projection_360.000000deg", "projection_450.000000deg", "projection_540.000000deg"]
    with engine.create(inp, device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        nz, nx = ctx.detector_shape
        batches, hpt, total = ctx.reference_shape()
        det = ctx.host_table("detector_data", "<f4")
        norm = 0.01 * float(det[19]) * float(det[20]) / total
        seed = 42
        for p, name in enumerate(files):
            img, _ = T.track(p, seed, 0, batches, hpt, ol.MATH_PORTABLE, n_threads=4)
            want = np.flip((img.reshape(4, nz, nx).astype(np.float64) * norm).transpose(1, 2, 0), axis=0)
            got = _read_like_reference(tmp_path / name, nz, nx, np.float64)
            assert np.allclose(got, want, rtol=0, atol=0.51e-8)  # "%.8lf" text of exactly the oracle's integers
            assert got.sum() > 0
            seed = engine.advance_seed(1, total, seed)  # GPU-build seed stepping between projections (MC-GPU_v1.3.cu:869)


def test_executable_reports_errors_like_the_reference(engine, tmp_path):
    res = subprocess.run([str(engine.EXE_PATH), str(tmp_path / "missing.in")], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and re.search("(?i)error", res.stdout)
    res = subprocess.run([str(engine.EXE_PATH)], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "ERROR" in res.stdout


def test_mcsimulation_mirror_runs_air_and_object_scans(engine, tmp_path):
    """MCSimulation.run_simulation flow (sim.py:370-427): air scan + object scan -> log(air/object) is a sane attenuation image."""
    mats, spc = cases.material_files(), cases.spectrum_file()
    kw = dict(n_projections=1, n_detector_pixels=(231, 96), detector_size=(717.024, 297.984))
    air = cases.simulation.MCSimulation(cases.geometry.MCAirGeometry(), mats, spc, n_histories=4_000_000, **kw)
    obj = cases.simulation.MCSimulation(cases.geometry.MCBoxGeometry(shape=(20, 20, 20), image_spacing=(10.0, 10.0, 10.0)), mats, spc,
                                        n_histories=4_000_000, **kw)
    (name_a, img_a, _, n_a), = air.run_simulation(tmp_path / "air", engine, mode="fast")
    (name_o, img_o, _, n_o), = obj.run_simulation(tmp_path / "obj", engine, mode="fast")
    assert n_a == n_o == 4_000_000
    nz, nx = img_a.shape[1:]
    a = _read_like_reference(name_a, nz, nx).sum(axis=-1)[:, :128]
    o = _read_like_reference(name_o, nz, nx).sum(axis=-1)[:, :128]
    centre = (slice(nz // 2 - 8, nz // 2 + 8), slice(40, 100))
    mu_l = np.log(a[centre].mean() / o[centre].mean())
    # 20 cm of water at ~60 keV effective energy: mu ~ 0.2/cm, minus scatter build-up
    assert 2.5 < mu_l < 4.5
