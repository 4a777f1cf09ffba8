"""CPU tests: the product's host model (C++ parsers/table builders behind the C ABI, device_id = -1)
against golden vectors dumped from the reference's read_input / load_voxels / load_material /
set_CT_trajectory / init_energy_spectrum / report_image (oracle/gen_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

import cases
import golden_util as gu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


@pytest.mark.parametrize("name", list(cases.CASES))
def test_tables_bit_identical_to_reference(name, case_dir, engine):
    g = gu.load(f"case_{name}.npz")
    with engine.create(case_dir(name), device=-1) as ctx:
        sc = gu.scalars(g)
        nproj = int(sc["num_projections"])
        assert ctx.num_projections == nproj
        for k in ("total_histories", "seed", "gpu_id", "threads_per_block", "histories_per_thread", "enable_specific_angles"):
            assert ctx.geti(k) == int(sc[k]), k
        assert np.float32(ctx.getf("mean_energy_spectrum")) == np.float32(sc["mean_energy_spectrum"])
        if nproj > 1:
            for k in ("D_angle", "initial_angle", "angularROI_0", "angularROI_1", "SRotAxisD", "vertical_translation"):
                assert ctx.getf(k) == sc[k], k
        # per-projection source / detector structs: byte for byte (80 B / 100 B each)
        assert np.array_equal(ctx.host_table("source_data"), g["source_data"])
        assert np.array_equal(ctx.host_table("detector_data"), g["detector_data"])
        # spectrum + Walker alias tables
        nb = int(g["num_bins_espc"])
        assert ctx.geti("num_spectrum_bins") == nb
        assert np.array_equal(ctx.host_table("espc", "<f4")[: nb + 1].view(np.uint32), g["espc"].view(np.uint32))
        assert np.array_equal(ctx.host_table("espc_cutoff", "<f4")[:nb].view(np.uint32), g["espc_cutoff"].view(np.uint32))
        assert np.array_equal(ctx.host_table("espc_alias", "<i2")[:nb], g["espc_alias"])
        # voxel header
        assert [ctx.geti(f"num_voxels_{a}") for a in "xyz"] == list(g["num_voxels"])
        assert np.array_equal(ctx.host_table("inv_voxel_size", "<f4"), g["inv_voxel_size"])
        assert np.array_equal(ctx.host_table("size_bbox", "<f4"), g["size_bbox"])
        dm, dm_ref = ctx.host_table("density_max", "<f4"), g["density_max"].copy()
        if dm[0] < 0:  # material 1 absent from the voxels: the reference overwrites its slot with 0.01*nominal (MC-GPU_v1.3.cu:2229-2230)
            assert dm_ref[0] == np.float32(0.01) * g["density_nominal"][0]
            dm_ref[0] = dm[0]
        assert np.array_equal(dm, dm_ref)
        used = g["used_materials"]
        assert np.array_equal(np.flatnonzero(ctx.host_table("noscco", "<i4")), used)
        assert np.array_equal(ctx.host_table("noscco", "<i4"), g["noscco"])
        assert np.array_equal(ctx.host_table("density_nominal", "<f4")[used], g["density_nominal"][used])
        assert (np.float32(ctx.getf("e0")), np.float32(ctx.getf("ide"))) == tuple(g["e0_ide"])
        nv = int(g["num_values"])
        assert ctx.geti("num_energy_values") == nv
        A = ctx.host_table("mfp_a", "<f4").reshape(nv, 25, 3)[:, used]
        B = ctx.host_table("mfp_b", "<f4").reshape(nv, 25, 3)[:, used]
        W = ctx.host_table("mfp_woodcock", "<f4").reshape(nv, 2)
        rows = g["sample_rows"]
        assert np.array_equal(A[rows].view(np.uint32), g["mfp_a_rows"].view(np.uint32))
        assert np.array_equal(B[rows].view(np.uint32), g["mfp_b_rows"].view(np.uint32))
        assert np.array_equal(W[rows[rows < nv - 1]].view(np.uint32), g["woodcock_rows"].view(np.uint32))
        # the reference leaves the last Woodcock entry uninitialised; ours repeats the previous slope
        assert W[nv - 1, 1] == W[nv - 2, 1] and np.isfinite(W[nv - 1]).all()
        mine = {
            "voxel_mat_dens": sha(ctx.host_table("voxel_mat_dens", "<f4")), "mfp_a_used": sha(A), "mfp_b_used": sha(B),
            "woodcock_but_last": sha(W[: nv - 1]),
            "pmax_used": sha(ctx.host_table("pmax", "<f4").reshape(-1, 25)[:nv, used]),
            "rayleigh_used": sha(np.stack([ctx.host_table(k, "<f4").reshape(25, 128)[used] for k in ("xco", "pco", "aco", "bco")])),
            "itl_itu_used": sha(np.stack([ctx.host_table(k).reshape(25, 128)[used] for k in ("itlco", "ituco")])),
            "compton_used": sha(np.stack([ctx.host_table(k, "<f4").reshape(40, 25)[:, used] for k in ("fco", "uico", "fj0")])),
        }
        for k, v in zip(g["digest_names"], g["digest_values"]):
            assert mine[str(k)] == str(v), f"table {k} differs from the reference"
        # output file names (float32 angle, MC-GPU_v1.3.cu:2787-2803)
        names = [os.path.basename(ctx.projection_file_name(p)) for p in range(nproj)]
        assert names == [str(s) for s in g["file_names"]]
        assert all(cases.simulation.PROJECTION_FILE_PATTERN.match(n) for n in names)


@pytest.mark.parametrize("name", ["catphan64_ct", "air"])
def test_ascii_projection_writer_matches_report_image(name, case_dir, engine, tmp_path):
    """Data lines byte-identical to the reference's fprintf("%.8lf ..."); same line count; np.loadtxt-able."""
    g = gu.load(f"case_{name}.npz")
    nb, hpt = [int(v) for v in g["nbatch_hpt"]]
    with engine.create(case_dir(name), device=-1) as ctx:
        p = ctx.num_projections - 1
        img = gu.dense(g, "ref", p, ctx.image_words)
        out = tmp_path / "proj"
        ctx.write_projection(p, img, nb * hpt, 1.0, str(out))
        lines = out.read_text().split("\n")
        assert len(lines) == int(g["ascii_num_lines"])
        data = [l for l in lines if not l.startswith("#")]
        want = [str(s) for s in g["ascii_first_rows"]]
        assert data[: len(want)] == want
        comments = [l for l in lines if l.startswith("#")]
        assert len(comments) == 20 + 5 + 1 - 1 + 0 or len(comments) >= 25
        assert comments[-3:] == [str(s) for s in g["ascii_comment_tail"]][-3:]  # histories / time / speed footer
        # the reference consumer (cbctmc/mc/projection.py:42-47)
        nz, nx = ctx.detector_shape
        arr = np.loadtxt(out, dtype=np.float64).reshape(nz, nx, 4)
        norm = (1.0 / 100.0) * float(np.float32(g["detector_data"].view("<f4")[19])) * float(np.float32(g["detector_data"].view("<f4")[20])) / (nb * hpt)
        back = np.rint(arr / norm).astype(np.uint64)
        assert np.array_equal(back.transpose(2, 0, 1).reshape(-1), img)


def test_bench_size_cirs_tables_equal_the_reference_parse(engine, tmp_path):
    """The bench-size CIRS workload as bench.py loads it (binary sidecar geometry.voxbin) against what the reference's own
    load_voxels / load_material made of the TEXT files (tests/golden/fullsize_ref_pin.json, oracle/gen_fullsize_pin.py):
    closes the common-mode path of the GPU parity tests, which feed the oracle the engine's host tables.  The 512^3 Catphan
    and the thorax are checked the same way in tests/test_gpu_fullsize.py (their fixtures take minutes to write)."""
    import bench
    import golden_util as gu
    inp = bench.build_workload(tmp_path, "cirs", int(1e8), 894, engine)
    assert (tmp_path / "geometry.voxbin").exists()
    with engine.create(inp, device=-1) as ctx:
        got, used = gu.fullsize_digests(ctx)
    pin = gu.fullsize_pin("cirs")
    assert used == pin["used_materials"]
    assert got == pin["sha256"]


def test_s0_bounds_bracket_the_reference_arithmetic(engine, case_dir):
    """The COMPAT kernel decides the Compton angle test from bounds lo <= S0 <= hi (model_device.cpp: build_s0_bounds) and rejects
    without a pass over the shells when xi lo > hi T(tau), which also needs S(theta) <= hi.  Both facts are held here against
    the reference's own float arithmetic (the oracle's, libm and portable math) for every material: at random energies, at the
    energies next to every bin edge, and for random deflections; the smallest slack is reported with the assertion."""
    import ctypes as C
    import oracle_lib as ol
    import parity
    with engine.create(case_dir("tissue22"), device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        raw = ctx.host_table("s0_bounds", "<f4")
        nb = (raw.size - 2) // (2 * 25)
        emin, inv_w = float(raw[-2]), float(raw[-1])
        b = raw[:-2].reshape(25, nb, 2)
        nosc = ctx.host_table("noscco", "<i4")
        e_hi = float(np.float32(ctx.getf("e0"))) + (ctx.geti("num_energy_values") - 1) / float(np.float32(ctx.getf("ide")))
        lib = ol.oracle()
        rng = np.random.default_rng(17)
        worst = 1.0
        for mat in [m for m in range(25) if nosc[m] > 0]:
            edges = emin + np.arange(0, nb + 1, 37) / inv_w
            energies = np.concatenate([rng.uniform(emin, e_hi, 400), edges, np.nextafter(edges.astype(np.float32), np.float32(0)),
                                       np.nextafter(edges.astype(np.float32), np.float32(1e9)), [emin, e_hi]]).astype(np.float32)
            energies = energies[(energies >= np.float32(emin)) & (energies <= np.float32(e_hi))]
            for E in energies:
                k = min(max(int(np.float32(np.float32(E - np.float32(emin)) * np.float32(inv_w))), 0), nb - 1)  # the kernel's bin
                lo, hi = float(b[mat, k, 0]), float(b[mat, k, 1])
                for mode in (ol.MATH_LIBM, ol.MATH_PORTABLE):
                    s0 = float(lib.oracle_compton_s(C.byref(T.ct), float(E), 2.0, mat, mode))
                    assert lo <= s0 <= hi, (mat, float(E), lo, s0, hi)
                    worst = min(worst, (s0 - lo) / max(s0, 1e-30) if lo > 0 else 1.0, (hi - s0) / max(s0, 1e-30))
                    cdt = float(np.float32(rng.uniform(0.0, 2.0)))
                    assert float(lib.oracle_compton_s(C.byref(T.ct), float(E), cdt, mat, mode)) <= hi, (mat, float(E), cdt)
        print("smallest relative slack of the S0 bounds:", worst)
        assert worst > 2e-5, worst  # the margin of 1e-4 is not eaten by the float arithmetic


@pytest.mark.parametrize("name", ["catphan64", "thorax64", "tissue22", "graded_u16"])
def test_coarse_woodcock_majorant_never_exceeds_the_table(name, case_dir, engine):
    """The FAST kernel flies with the Woodcock mean free path of a COARSE energy bin (64 table bins, LDS: LdsLayout::wood) instead of
    interpolating the reference's table (MC-GPU_kernel_v1.3.cu:228).  Delta tracking is exact for any majorant -- i.e. for any mean
    free path that is NOWHERE larger than the table's: checked here at both ends of every table bin (the table is linear in
    between), and that it is not wastefully small either (within 12 % of the bin's own minimum above 20 keV)."""
    with engine.create(case_dir(name), device=-1) as ctx:
        nv = ctx.geti("num_energy_values")
        w = ctx.host_table("mfp_woodcock", "<f4").reshape(nv, 2).astype(np.float64)
        coarse = ctx.host_table("woodcock_coarse", "<f4").astype(np.float64)
        e0, ide = ctx.getf("e0"), ctx.getf("ide")
    assert coarse.size == (nv + 63) // 64
    i = np.arange(nv)
    lo_end, hi_end = w[:, 0] + (e0 + i / ide) * w[:, 1], w[:, 0] + (e0 + (i + 1) / ide) * w[:, 1]
    fine_min = np.minimum(lo_end, hi_end)
    c = coarse[i >> 6]
    assert np.all(c > 0) and np.all(c <= fine_min * (1.0 - 5e-7)), float((c / fine_min).max())
    e = e0 + i / ide
    assert (c / fine_min)[e > 20000.0].min() > 0.88
