"""Dose tallies: tally_materials_dose / tally_voxel_energy_deposition and their reports.

CPU: the oracle against the fixture captured from the reference build (integer tallies, bit for bit), the engine's
report writers against the reference's report files (data lines, binary volumes, table rows).
GPU: the COMPAT kernel bit-exact against the oracle (portable math), the FAST kernel statistically.
"""
import hashlib

import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
import parity

CASE = "catphan64_dose"


def _golden_dose(g, tag):
    shape = tuple(int(v) for v in g["dose_voxels_shape"])
    vox = np.zeros(int(np.prod(shape)), dtype=np.uint64)
    vox[g[f"dose_voxels_{tag}_idx"]] = g[f"dose_voxels_{tag}_val"]
    return vox.reshape(shape)


def test_dose_section_is_parsed_and_clipped(engine, case_dir, tmp_path):
    with engine.create(case_dir(CASE), device=-1) as ctx:
        flags, roi, shape = ctx.dose_info()
        assert flags == 3 and roi == [8, 55, 4, 59, 16, 47] and shape == (32, 56, 48)
    with engine.create(case_dir("catphan64"), device=-1) as ctx:
        assert ctx.dose_info()[0] == 0  # the reference template's default: NO / NO
    # an ROI larger than the volume is clipped to it (load_voxels, MC-GPU_v1.3.cu:2058-2064)
    with engine.create(case_dir(CASE, dose_roi=((1, 500), (60, 64), (1, 64))), device=-1) as ctx:
        assert ctx.dose_info()[1] == [0, 63, 59, 63, 0, 63]
    # an inverted ROI is an input error (MC-GPU_v1.3.cu:1677-1684)
    with pytest.raises(engine.EngineError) as e:
        engine.create(case_dir(CASE, dose_roi=((9, 3), (1, 4), (1, 4))), device=-1)
    assert e.value.code == -2 and "ERROR" in e.value.message


def test_oracle_dose_matches_reference_fixture(engine, case_dir):
    g = gu.load(f"case_{CASE}.npz")
    nb, hpt = [int(v) for v in g["nbatch_hpt"]]
    with engine.create(case_dir(CASE), device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        roi = ctx.dose_info()[1]
        assert roi == [int(v) for v in g["dose_roi"]]
        for mode, tag, want_mat in ((ol.MATH_LIBM, "libm", g["dose_materials_ref"]), (ol.MATH_PORTABLE, "portable", g["dose_materials_portable"])):
            vox, mat = T.enable_dose(roi, True)
            for p in range(ctx.num_projections):
                T.track(p, 42 + 1000 * p, 0, nb, hpt, mode, n_threads=4)
            want_vox = _golden_dose(g, tag)
            budget = 0 if (tag == "portable" or ol.reference_available()) else 60  # a foreign libm may flip a few histories
            assert np.count_nonzero(vox != want_vox) <= budget, tag
            assert np.count_nonzero(mat != want_mat) <= (0 if budget == 0 else 8), tag
            # the voxel tally over the whole ROI and the material tally see the same deposits inside the ROI
            assert vox[..., 0].sum() <= mat[:, 0].sum()


def test_dose_reports_match_reference_files(engine, case_dir, tmp_path):
    """report_voxels_dose / report_materials_dose on the reference's own tallies: data lines, .raw volumes, table rows."""
    g = gu.load(f"case_{CASE}.npz")
    nb, hpt = [int(v) for v in g["nbatch_hpt"]]
    inp = case_dir(CASE)
    with engine.create(inp, device=-1) as ctx:
        text = ctx.write_dose_report(_golden_dose(g, "libm"), g["dose_materials_ref"], nb * hpt, seconds=1.0)
    dose_file = inp.parent / "dose.dat"
    lines = dose_file.read_text().split("\n")
    sep = max(i for i, l in enumerate(lines) if l.startswith("# ====="))
    assert lines[sep + 1:] == [str(s) for s in g["dose_file_body"]]
    got_sha = [hashlib.sha256((inp.parent / ("dose.dat" + sfx)).read_bytes()).hexdigest() for sfx in (".raw", "_2sigma.raw")]
    assert got_sha == [str(s) for s in g["dose_raw_sha256"]]
    rows = [l for l in text.split("\n") if l.startswith("\t")]
    assert rows == [str(s) for s in g["dose_stdout_rows"]]
    assert "VOXEL ROI DOSE TALLY REPORT" in text and "MATERIALS TOTAL DOSE TALLY REPORT" in text


@pytest.mark.gpu
def test_compat_dose_bit_exact_vs_oracle(engine, case_dir):
    nb, hpt = 256, 150
    with engine.create(case_dir(CASE), device=0) as ctx:
        T = parity.tables_from_context(ctx)
        roi = ctx.dose_info()[1]
        vox_cpu, mat_cpu = T.enable_dose(roi, True)
        for p in range(ctx.num_projections):
            seed = 42 + 1000 * p
            img_gpu, _, _ = ctx.run_projection(p, nb, mode="compat", seed=seed, hpt=hpt)
            img_cpu, _ = T.track(p, seed, 0, nb, hpt, ol.MATH_PORTABLE, n_threads=4)
            assert np.array_equal(img_gpu.reshape(-1), img_cpu)
        vox_gpu, mat_gpu = ctx.dose_read()
        assert mat_gpu[:, 0].sum() > 0
        assert np.array_equal(mat_gpu, mat_cpu)
        assert np.array_equal(vox_gpu, vox_cpu)
        ctx.dose_clear()
        vox0, mat0 = ctx.dose_read()
        assert not vox0.any() and not mat0.any()


@pytest.mark.gpu
def test_fast_dose_statistics_vs_oracle(engine, case_dir):
    """FAST kernel: energy deposited per history, per material and in coarse voxel blocks, against the oracle sample."""
    nb, hpt = 3000, 150
    n_gpu = 6_000_000
    with engine.create(case_dir(CASE), device=0) as ctx:
        T = parity.tables_from_context(ctx)
        roi = ctx.dose_info()[1]
        vox_cpu, mat_cpu = T.enable_dose(roi, True)
        T.track(0, 42, 0, nb, hpt, ol.MATH_LIBM, n_threads=8)
        ctx.run_projection(0, n_gpu, mode="fast", seed=7)
        vox_gpu, mat_gpu = ctx.dose_read()
        n_cpu = nb * hpt
        # deterministic: a second run on cleared tallies gives the same integers (per-workgroup LDS accumulators included)
        ctx.dose_clear()
        ctx.run_projection(0, n_gpu, mode="fast", seed=7)
        vox2, mat2 = ctx.dose_read()
        assert np.array_equal(vox2, vox_gpu) and np.array_equal(mat2, mat_gpu)
        # per material: mean deposited energy per history; sigma from the oracle's own <E^2> tally
        for m in np.flatnonzero(mat_cpu[:, 0]):
            mean_cpu = mat_cpu[m, 0] / 100.0 / n_cpu
            mean_gpu = mat_gpu[m, 0] / 100.0 / n_gpu
            var = max(mat_cpu[m, 1] / n_cpu - mean_cpu ** 2, 0.0) / n_cpu
            if mat_cpu[m, 0] / 100.0 / 3.0e4 < 200:  # fewer than ~200 deposits: too noisy to test
                continue
            assert abs(mean_gpu - mean_cpu) < 4.0 * np.sqrt(var) + 2e-3 * mean_cpu, f"material {m + 1}: {mean_gpu} vs {mean_cpu} eV/history"
        # 8x8x8 voxel blocks
        def blocks(v):
            d = v[..., 0].astype(np.float64)
            z, y, x = (s // 8 * 8 for s in d.shape)
            return d[:z, :y, :x].reshape(z // 8, 8, y // 8, 8, x // 8, 8).sum(axis=(1, 3, 5))
        bc, bg = blocks(vox_cpu) / n_cpu, blocks(vox_gpu) / n_gpu
        counts = blocks(vox_cpu) / 3.0e6  # ~ number of deposits (30 keV * 100 each)
        mask = counts > 100
        assert mask.sum() > 20
        z = (bg[mask] - bc[mask]) / (bc[mask] * np.sqrt(2.0 / counts[mask]))
        assert np.mean(np.abs(z) > 3.0) < 0.03 and np.abs(z).max() < 6.0 and abs(z.mean()) < 0.5
