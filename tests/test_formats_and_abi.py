"""CPU tests: wire-format edge cases, error behaviour and the C-ABI surface (no GPU needed)."""
import gzip
import os
import re
from pathlib import Path

import numpy as np
import pytest

import cases

ROOT = Path(__file__).resolve().parents[1]


def test_abi_exports_every_declared_symbol(engine):
    header = (ROOT / "include" / "mcgpu_amd.h").read_text()
    declared = sorted(set(re.findall(r"\b(mcgpu_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 19
    lib = engine.load_library()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/mcgpu_amd.h but not exported"
    assert sorted(engine.ABI_SYMBOLS) == declared
    assert lib.mcgpu_abi_version() == 1


def test_no_torch_types_and_no_oracle_in_product():
    """The ABI is plain C; the product tree never references the test oracle."""
    header = (ROOT / "include" / "mcgpu_amd.h").read_text()
    assert "torch" not in header and "at::" not in header
    for f in list((ROOT / "4d-cbct-mc_amd").rglob("*.py")) + list((ROOT / "4d-cbct-mc_amd" / "csrc").glob("*.[chi]*")):
        text = f.read_text()
        assert "liboracle" not in text and "oracle_lib" not in text and "mcgpu_oracle.h" not in text, f


def test_errors_are_reported_not_thrown(engine, tmp_path):
    with pytest.raises(engine.EngineError) as e:
        engine.create(tmp_path / "does_not_exist.in", device=-1)
    assert e.value.code == -1 and "ERROR" in e.value.message
    bad = tmp_path / "bad.in"
    bad.write_text("#[SECTION SIMULATION CONFIG v.2009-05-12]\n1000\n42\n0\n100\n150\n")  # 100 threads: not a multiple of 32
    with pytest.raises(engine.EngineError) as e:
        engine.create(bad, device=-1)
    assert e.value.code == -2 and re.search("(?i)error", e.value.message)


def test_host_only_context_refuses_to_launch(engine, case_dir):
    """No GPU, no fallback: a context without a device cannot run the hot path."""
    with engine.create(case_dir("air"), device=-1) as ctx:
        with pytest.raises(engine.EngineError) as e:
            ctx.run_projection(0, 1000, mode="fast")
        assert "ERROR" in e.value.message


def test_launch_shape_matches_reference_examples(engine):
    # N=1e8 -> 5209 blocks x 128 x 150; N=11 903 320 312 -> 65000 x 128 x 1431
    assert engine.launch_shape(100_000_000, 128, 150) == (5209, 150, 100_012_800)
    assert engine.launch_shape(11_903_320_312, 128, 150) == (65000, 1431, 11_905_920_000)
    assert engine.launch_shape(1, 128, 150) == (1, 150, 19200)
    assert engine.advance_seed(1, 100_012_800, 42) == 1267439713
    assert engine.advance_seed(0, 5, 77) == 77


def _write_vox(path, nx, ny, nz, body, gz=False, header_extra=""):
    text = (f"# comment\n{header_extra}[SECTION VOXELS HEADER v.2008-04-13]\n{nx} {ny} {nz}  # SIZE\n0.5 0.25 1.0  # SPACING\n1\n2\n1\n"
            "[END OF VXH SECTION]\n#\n" + body)
    if gz:
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        Path(path).write_text(text)


def _input_for(tmp_path, vox, **kw):
    sim_kw = dict(n_projections=1, n_histories=1000, n_detector_pixels=(8, 4), detector_size=(717.024, 297.984))
    sim_kw.update(kw)
    text = cases.simulation.create_mcgpu_input(vox, cases.material_files(), cases.spectrum_file(), (10.0, -900.0, 10.0), tmp_path, **sim_kw)
    p = tmp_path / "input.in"
    p.write_text(text)
    return p


def test_voxel_parser_edge_cases(engine, tmp_path):
    # blank lines, comment lines, leading blanks, exponent notation, gzip; x fastest
    body = "1 0.001300\n 6 1.000000\n\n\n# a comment\n6 1.5e0\n21 2.160000\n\n6   0.9999999\n1 0.0013\n\n\n"
    vox = tmp_path / "g.vox.gz"
    _write_vox(vox, 2, 3, 1, body, gz=True)
    with engine.create(_input_for(tmp_path, vox), device=-1) as ctx:
        md = ctx.host_table("voxel_mat_dens", "<f4").reshape(-1, 2)
        assert np.array_equal(md[:, 0], np.float32([1, 6, 6, 21, 6, 1]) + np.float32(0.0001))
        assert np.array_equal(md[:, 1], np.float32([0.0013, 1.0, 1.5, 2.16, 0.9999999, 0.0013]))
        assert np.array_equal(ctx.host_table("size_bbox", "<f4"), np.float32([2 * 0.5, 3 * 0.25, 1.0]))
        dm = ctx.host_table("density_max", "<f4")
        assert dm[0] == np.float32(0.0013) and dm[5] == np.float32(1.5) and dm[20] == np.float32(2.16) and dm[1] == -999.0


@pytest.mark.parametrize("body,needle", [
    ("0 1.0\n1 1.0\n", "zero or negative"), ("1 0.0\n1 1.0\n", "density"), ("26 1.0\n1 1.0\n", "too high"), ("1 1.0\n", "ends after"),
])
def test_voxel_parser_rejects_bad_data(engine, tmp_path, body, needle):
    vox = tmp_path / "g.vox"
    _write_vox(vox, 2, 1, 1, body)
    with pytest.raises(engine.EngineError) as e:
        engine.create(_input_for(tmp_path, vox), device=-1)
    assert e.value.code == -2 and needle in e.value.message and "ERROR" in e.value.message


def test_voxel_writers_agree_and_round_trip(engine, tmp_path):
    rng = np.random.default_rng(5)
    mats = rng.choice([1, 6, 21], size=(7, 5, 3)).astype(np.uint8)
    dens = np.where(mats == 1, 0.0013, np.where(mats == 6, rng.uniform(0.9, 1.1, mats.shape), 2.16)).astype(np.float32)
    geo = cases.geometry.MCGeometry(mats, dens, (2.0, 3.0, 4.0))
    geo.save_mcgpu_geometry(tmp_path / "py.vox", compress=False)
    geo.save_mcgpu_geometry(tmp_path / "cc.vox", compress=False, engine=engine)
    geo.save_mcgpu_geometry(tmp_path / "cc.vox.gz", compress=True, engine=engine)
    body = lambda p: [l for l in Path(p).read_text().split("\n") if l and not l.startswith("#") and "[" not in l and not l.endswith("(1=YES, 0=NO)")]
    py, cc = body(tmp_path / "py.vox"), body(tmp_path / "cc.vox")
    assert py[3:] == cc[3:]  # voxel lines identical ("<mat> <density:.6f>")
    assert gzip.open(tmp_path / "cc.vox.gz", "rt").read() == (tmp_path / "cc.vox").read_text()
    with engine.create(_input_for(tmp_path, tmp_path / "cc.vox.gz"), device=-1) as ctx:
        m_w, d_w, sp = geo.mcgpu_arrays()  # rot90(k=3) + swapped x/y spacing (geo.py:589-599)
        assert [ctx.geti(f"num_voxels_{a}") for a in "xyz"] == list(m_w.shape)
        md = ctx.host_table("voxel_mat_dens", "<f4").reshape(m_w.shape[2], m_w.shape[1], m_w.shape[0], 2)
        assert np.array_equal(md[..., 0], (m_w.transpose(2, 1, 0).astype(np.float32) + np.float32(0.0001)))
        want = np.float32([float(f"{v:.6f}") for v in d_w.transpose(2, 1, 0).reshape(-1)]).reshape(md.shape[:3])
        assert np.array_equal(md[..., 1], want)
        assert np.allclose(ctx.host_table("voxel_size", "<f4"), [0.3, 0.2, 0.4])


def test_input_parser_variants(engine, tmp_path):
    vox = tmp_path / "g.vox"
    _write_vox(vox, 1, 1, 1, "1 0.0013\n")
    # explicit angles override the projection count; first angle defines the initial angle
    with engine.create(_input_for(tmp_path, vox, projection_angles=[10.0, 200.0, 359.99999]), device=-1) as ctx:
        assert ctx.num_projections == 3 and ctx.geti("enable_specific_angles") == 1
        assert abs(ctx.getf("initial_angle") - np.deg2rad(10.0)) < 1e-12
        names = [ctx.projection_file_name(p).split("_")[-1] for p in range(3)]
        assert names == ["010.000000deg", "200.000000deg", "360.000000deg"]
    # zero projections behaves like one (MC-GPU_v1.3.cu:1539-1540)
    with engine.create(_input_for(tmp_path, vox, n_projections=0), device=-1) as ctx:
        assert ctx.num_projections == 1
        det = ctx.host_table("detector_data", "<i4")
        assert det[-1] == 0  # beam along +Y, single projection: detector not rotated
    # pencil beam
    with engine.create(_input_for(tmp_path, vox, source_polar_aperture=(0.0, 0.0), source_azimuthal_aperture=0.0), device=-1) as ctx:
        src = ctx.host_table("source_data", "<f4")
        assert src[15] == 0.0 and src[17] == 0.0 and src[18] == 0.0 and src[19] == 0.0  # cos_theta_low, D_cos_theta, D_phi, max_height
    with pytest.raises(engine.EngineError):
        engine.create(_input_for(tmp_path, vox, n_projections=2000), device=-1)


def test_catphan_recipe_and_padding():
    g = cases.geometry.MCCatPhan604Geometry(shape=(100, 100, 100), scale=0.2)
    nums = {cases.materials.material_number(k) for k in ("air", "pmp", "ldpe", "h2o", "polystyrene", "bone_020", "acrylic", "bone_050", "delrin", "teflon")}
    assert set(np.unique(g.materials)) == nums
    assert g.materials[50, 50, 50] == cases.materials.material_number("h2o")
    assert g.materials[0, 0, 0] == 1 and g.densities[0, 0, 0] == np.float32(0.0013)
    p = g.pad_to_shape((104, 100, 108))
    assert p.image_shape == (104, 100, 108) and np.array_equal(p.materials[2:102, :, 4:104], g.materials)
    assert cases.simulation.source_position_for((305.0, 300.0, 152.0)) == (152.5, -850.0, 76.0)


def test_thorax_bone_texture_follows_the_reference_bone_mapper():
    """`MCThoraxLikeGeometry(bone_texture=True)`: the rule of the reference's BoneMaterialMapper (geo.py:138-166) applied to a seeded HU
    field inside the bones -- all four bone classes occur side by side, bone_100 only on the one-voxel outline of the bone
    segmentation, every class at its nominal density -- and of its AirMaterialMapper to the lungs; nothing else changes, the same
    volume for the same seed."""
    from scipy import ndimage
    g = cases.pkg.geometry
    shape = (128, 128, 64)
    smooth = g.MCThoraxLikeGeometry(shape=shape, image_spacing=(4.0, 4.0, 4.0))
    tex = g.MCThoraxLikeGeometry(shape=shape, image_spacing=(4.0, 4.0, 4.0), bone_texture=True)
    again = g.MCThoraxLikeGeometry(shape=shape, image_spacing=(4.0, 4.0, 4.0), bone_texture=True)
    assert np.array_equal(tex.materials, again.materials) and np.array_equal(tex.densities, again.densities)
    classes = [g.material_number(k) for k in ("red_marrow", "bone_020", "bone_050", "bone_100")]
    bone = np.isin(smooth.materials, classes)
    assert np.array_equal(np.isin(tex.materials, classes), bone)                       # the segmentation itself is unchanged
    lung, air = smooth.materials == g.material_number("lung"), g.material_number("air")
    assert np.array_equal(tex.materials[~bone & ~lung], smooth.materials[~bone & ~lung]) and (tex.materials[bone] != smooth.materials[bone]).mean() > 0.3
    # the lungs (the reference's AirMaterialMapper, geo.py:168-183): some voxels become air, nothing else happens there
    changed = lung & (tex.materials != smooth.materials)
    assert np.all(tex.materials[changed] == air) and np.all(tex.densities[changed] == np.float32(g.MATERIALS_125KEV["air"])) and 0.05 < changed.sum() / lung.sum() < 0.25
    for k, ident in zip(classes, ("red_marrow", "bone_020", "bone_050", "bone_100")):
        m = tex.materials == k
        assert m.sum() > 50 and np.all(tex.densities[m] == np.float32(g.MATERIALS_125KEV[ident]))
    outline = bone & ~ndimage.binary_erosion(bone)
    assert not np.any((tex.materials == classes[3]) & ~outline)


def test_binary_voxel_sidecar_equals_text_parse(engine, tmp_path):
    """geometry.voxbin yields exactly the host model of the text file it shadows."""
    rng = np.random.default_rng(4)
    g = cases.geometry.MCBoxGeometry(shape=(14, 9, 11), image_spacing=(7.0, 9.0, 11.0), material="h2o")
    g.materials[3:9, 2:6, 4:8] = cases.materials.material_number("bone_050")
    g.densities[:] = rng.uniform(0.001, 2.2, g.densities.shape).astype(np.float32)  # more digits than "%.6f" keeps
    sim_kw = dict(n_projections=1, n_histories=1000, **cases.SMALL_DET)
    a, b = tmp_path / "text", tmp_path / "bin"
    for folder, side in ((a, False), (b, True)):
        sim = cases.simulation.MCSimulation(g, cases.material_files(), cases.spectrum_file(), **sim_kw)
        sim.prepare_simulation(folder, compress_geometry=True, engine=engine, binary_sidecar=side)
    assert (b / "geometry.voxbin").exists() and not (a / "geometry.voxbin").exists()
    with engine.create(a / "input.in", device=-1) as ta, engine.create(b / "input.in", device=-1) as tb:
        for name in ("voxel_mat_dens", "density_max", "mfp_woodcock", "size_bbox", "inv_voxel_size"):
            assert np.array_equal(ta.host_table(name), tb.host_table(name)), name
    # the sidecar is what gets read: corrupting the text body goes unnoticed, a stale sidecar is ignored
    text = gzip.open(b / "geometry.vox.gz", "rt").read()
    import os, time
    with gzip.open(b / "geometry.vox.gz", "wt") as f:
        f.write(text.replace("2 1.4", "9 1.4", 1) if "2 1.4" in text else text[: len(text) // 2])
    os.utime(b / "geometry.voxbin", (time.time() + 5, time.time() + 5))
    with engine.create(b / "input.in", device=-1) as tb2, engine.create(a / "input.in", device=-1) as ta2:
        assert np.array_equal(tb2.host_table("voxel_mat_dens"), ta2.host_table("voxel_mat_dens"))
    (b / "geometry.voxbin").write_bytes(b"not a sidecar")
    with pytest.raises(engine.EngineError) as e:
        engine.create(b / "input.in", device=-1)
    assert "ERROR" in e.value.message


def test_aluminium_table_with_float_formatted_integer_columns(engine, tmp_path):
    """The reference's aluminium table writes ITL/ITU and KZCO/KSCO as "1.0 4.0".  The reference reads them with an
    unchecked sscanf("%d %d") that stops at the '.', leaving ITU uninitialised (MC-GPU_v1.3.cu:2387-2392); the engine
    reads the numbers that are written (DESIGN.md deviation 10): the file as shipped must give exactly the tables of a
    copy whose columns are spelled as integers -- which is also what the reference build reads from that copy
    (test_host_tables.py, case tissue22)."""
    raw_files = cases.material_files(raw_aluminium=True)
    k = cases.materials.material_number("aluminium") - 1
    text = raw_files[k].read_text()
    assert re.search(r"^0\.0 0\.0 \S+ \S+ 1\.0 4\.0$", text, re.M)
    g = cases.geometry.MCBoxGeometry(shape=(6, 6, 6), image_spacing=(10.0, 10.0, 10.0), material="aluminium")
    tabs = []
    for sub, files in (("raw", raw_files), ("int", cases.material_files())):
        sim = cases.simulation.MCSimulation(g, files, cases.spectrum_file(), n_projections=1, n_histories=1000, **cases.SMALL_DET)
        with engine.create(sim.prepare_simulation(tmp_path / sub), device=-1) as ctx:
            tabs.append({n: ctx.host_table(n).copy() for n in ("itlco", "ituco", "xco", "pco", "aco", "bco", "fco", "uico", "fj0", "noscco", "mfp_a")})
    for n in tabs[0]:
        assert np.array_equal(tabs[0][n], tabs[1][n]), n
    itu = tabs[0]["ituco"].reshape(25, 128)[k]
    assert itu[0] == 4 and itu.max() == 128 and tabs[0]["noscco"].view("<i4")[k] == 5


def test_ascii_writer_is_safe_under_concurrent_calls(engine, case_dir, tmp_path):
    """Two threads writing projection files through the public ABI at the same time (e.g. one scan per GPU in one process)
    give the bytes of the single-threaded writes: no buffer is shared between calls."""
    import threading
    with engine.create(case_dir("catphan64_ct"), device=-1) as ctx:
        rng = np.random.default_rng(3)
        nz, nx = ctx.detector_shape
        images = [rng.integers(0, 10**9, size=(4, nz, nx), dtype=np.uint64) for _ in range(4)]
        want = []
        for k, img in enumerate(images):
            f = tmp_path / f"serial_{k}"
            ctx.write_projection(k, img, 1_000_000 + k, 1.5, file_name=str(f))
            want.append(f.read_bytes())
        errors = []

        def worker(t):
            try:
                for rep in range(15):
                    for k, img in enumerate(images):
                        f = tmp_path / f"thread{t}_{rep}_{k}"
                        ctx.write_projection(k, img, 1_000_000 + k, 1.5, file_name=str(f))
                        if f.read_bytes() != want[k]:
                            errors.append((t, rep, k))
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:5]
        assert want[0].startswith(b"# \n#     ****") and b"Simulated x rays:    1000000\n" in want[0]


def test_mpirun_shim_translates_the_reference_command_line(tmp_path):
    """docker/mpirun is what `mpirun --tag-output -v -n <N> MC-GPU_v1.3.x <input>` (cbctmc/mc/simulation.py:187-198) resolves to
    inside the ROCm image: it must hand the executable its input file and the rank count as `--gpus N`."""
    import subprocess
    shim = ROOT / "docker" / "mpirun"
    probe = tmp_path / "probe.sh"
    probe.write_text('#!/bin/sh\nfor a in "$@"; do printf "[%s]" "$a"; done; echo\n')
    probe.chmod(0o755)
    run = lambda *argv: subprocess.run(["sh", str(shim), *argv], capture_output=True, text=True, check=True).stdout.strip()
    assert run("--tag-output", "-v", "-n", "2", str(probe), "/host/run/input.in") == "[/host/run/input.in][--gpus][2]"
    assert run("-np", "8", str(probe), "a b.in") == "[a b.in][--gpus][8]"       # spaces survive
    assert run(str(probe), "input.in") == "[input.in][--gpus][1]"                # no -n: one rank
    assert run("--tag-output", "-v", "-n", "4", "/bin/echo", "in") == "in --gpus 4"


def test_option_structs_carry_their_size(engine, case_dir):
    """mcgpu_scan_options / mcgpu_fdk_options begin with `struct_size`: a caller that forgot to set it is refused (its fields would
    otherwise be read at the wrong offsets), one that was compiled against a SHORTER struct gets the defaults of the fields it does
    not know -- here a scan-options struct cut before `shard`, handed to a host-only context, fails for the right reason (no
    device), not for a garbage shard mode."""
    import ctypes as C
    lib = engine.load_library()
    with engine.create(case_dir("air"), device=-1) as ctx:
        o = engine.ScanOptions()
        r = engine.ScanReport()
        assert lib.mcgpu_run_scan(ctx.h, C.byref(o), C.byref(r)) != 0
        assert b"struct_size" in lib.mcgpu_last_error()
        short = engine.ScanOptions.shard.offset
        o.struct_size = short
        o.shard, o.projection_stride, o.projection_phase = 77, -5, 99  # beyond struct_size: must not be looked at
        assert lib.mcgpu_run_scan(ctx.h, C.byref(o), C.byref(r)) != 0
        msg = lib.mcgpu_last_error().decode()
        assert "no device" in msg and "projection_phase" not in msg, msg
    recon = __import__("cases").pkg.reconstruction
    fo = recon._FdkOptions()
    lib.mcgpu_fdk_reconstruct.restype = C.c_int
    assert lib.mcgpu_fdk_reconstruct(C.byref(fo), None, None, None) != 0
    assert b"struct_size" in lib.mcgpu_last_error()


def test_clone_shares_the_parsed_model(engine, case_dir):
    """mcgpu_clone: a further context of the same simulation without parsing anything again (one per device in the
    executable's --gpus path); host tables and configuration are those of the parsed context, and it outlives it."""
    src = engine.create(case_dir("slab_angles"), device=-1)
    twin = src.clone(device=-1)
    names = ("voxel_mat_dens", "mfp_woodcock", "mfp_a", "source_data", "detector_data", "espc_alias", "noscco")
    want = {n: src.host_table(n).copy() for n in names}
    nproj, seed = src.num_projections, src.geti("seed")
    src.close()
    for n in names:
        assert np.array_equal(twin.host_table(n), want[n]), n
    assert twin.num_projections == nproj == 3 and twin.geti("seed") == seed
    assert twin.projection_file_name(1).endswith("projection_300.500000deg")
    twin.close()


def test_committed_pmc_summaries_belong_to_this_kernel_build():
    """bench.py quotes `roofline.traffic` from the committed rocprofv3 PMC summaries and refuses one that was collected from
    another build of the FAST kernel (source hash) or another workload.  A kernel change without a fresh collection
    (tools/collect_round.sh) would silently turn the driver's `traffic` into null: caught here instead."""
    import json
    import bench
    for workload, name in (("catphan", "pmc_summary_latest.json"), ("cirs", "pmc_summary_cirs.json"), ("thorax", "pmc_summary_thorax.json"),
                           ("thorax_textured", "pmc_summary_thorax_textured.json")):
        stamp = json.loads((ROOT / "profiles" / name).read_text())["_stamp"]
        assert stamp["workload"] == workload, name
        assert stamp["kernel_source_sha16"] == bench.kernel_source_hash(), f"{name} is of another kernel build: run tools/collect_round.sh on a GPU box"
        d, src = bench.pmc_summary(workload)
        assert d is not None and name in src
    roof, valu = bench.roofline_block("catphan", int(1e8), 4.0)
    assert roof["traffic"] > 1e9 and 0 < roof["frac"] < 1 and roof["bound"] == "hbm" and 0.3 < valu["lane_utilisation"] < 1
    roof, _ = bench.roofline_block("catphan", int(5e7), 2.0)  # the summaries are per 1e8-history launch
    assert roof["traffic"] is None and roof["frac"] > 0


def _decode_tile_record(rec, bit):
    """The FAST kernel's lookup of voxel `bit` in a tile record (csrc/track_pool.inc: flight_locate), restated: None = ask the volume."""
    ab, code, lo, hi = (int(x) for x in rec)
    mask = lo | (hi << 32)
    if code >= 0xC0000000:
        return None
    if not (mask >> bit) & 1:
        return ab & 0xFF
    if code == 0:
        return (ab >> 8) & 0xFF
    rank = bin(mask & ((1 << bit) - 1)).count("1")
    wide = code >> 30
    sel = (code >> (rank << wide)) & (1 + 2 * wide)
    return (ab >> (8 + 8 * sel)) & 0xFF


def test_tile_records_round_trip(engine):
    """csrc/device_model.hpp: encode_tile_record (one definition for the host builder and for the device builder of a warped volume)
    against the kernel's lookup restated above, on random tiles of one to six palette entries with and without padding: every voxel
    comes back, tiles of one or two entries keep the round-4 record (code 0, a = the first entry met), three entries are held in the
    record up to 30 minority voxels, four up to 15, everything else asks the volume."""
    rng = np.random.default_rng(11)
    tiles, kinds = [], []
    for n_entries in (1, 2, 3, 4, 5, 6):
        for majority in (0.3, 0.55, 0.8, 0.95):
            for _ in range(40):
                pal = rng.choice(256, size=n_entries, replace=False)
                p = np.full(n_entries, (1 - majority) / max(n_entries - 1, 1)); p[rng.integers(n_entries)] = majority if n_entries > 1 else 1.0
                t = pal[rng.choice(n_entries, size=64, p=p / p.sum())].astype(np.int16)
                if rng.random() < 0.3:  # an edge tile: the upper x / y / z part is padding
                    v = np.arange(64)
                    t[((v & 3) >= rng.integers(1, 5)) | (((v >> 2) & 3) >= rng.integers(1, 5)) | ((v >> 4) >= rng.integers(1, 5))] = -1
                tiles.append(t)
    tiles = np.array(tiles)
    recs = engine.kat_tile_records(tiles)
    held = {1: 0, 2: 0, 3: 0, 4: 0, 5: 0, 6: 0}
    for t, r in zip(tiles, recs):
        valid = t[t >= 0]
        vals, counts = np.unique(valid, return_counts=True)
        n = len(vals)
        others = len(valid) - counts.max() if n else 0
        expect_volume = n >= 5 or (n == 3 and others > 30) or (n == 4 and others > 15)
        assert (int(r[1]) >= 0xC0000000) == expect_volume, (n, others, hex(int(r[1])))
        if n <= 2:
            assert r[1] == 0 and (n == 0 or (int(r[0]) & 0xFF) == valid[0])
        if expect_volume:
            continue
        held[n] += 1
        for bit in np.nonzero(t >= 0)[0]:
            assert _decode_tile_record(r, int(bit)) == t[bit], (n, bit)
    assert held[3] > 50 and held[4] > 30 and held[5] == held[6] == 0


def test_environment_knobs_go_through_one_registry(engine, tmp_path):
    """csrc/knobs.cpp is the engine's only reader of the environment: every name the sources ask for is a row of the
    table; the table is what `MC-GPU_v1.3.x --knobs`, mcgpu_knob_table and INTEGRATION.md show; a misspelt MCGPU_* variable is
    reported once -- in words the reference's log scanner (cbctmc/mc/simulation.py:204: any-case "error") does not take for a failure."""
    import subprocess
    import sys
    csrc = ROOT / "4d-cbct-mc_amd" / "csrc"
    sources = [f for f in csrc.iterdir() if f.suffix in (".cpp", ".hpp", ".hip", ".inc")]
    reads = sum(len(re.findall(r"\bgetenv\s*\(", f.read_text())) for f in sources)
    assert reads == 1, "the one environment read lives in knobs.cpp"
    assert sum(f.read_text().count("getenv") for f in sources) <= 3
    table = {k["name"]: k for k in engine.knob_table()}
    asked = set()
    for f in sources:
        asked |= set(re.findall(r'knob_(?:str|set|int|float)\(\s*"(MCGPU_[A-Z0-9_]+)"', f.read_text()))
        asked |= set(re.findall(r'env_int\(\s*"(MCGPU_[A-Z0-9_]+)"', f.read_text()))
    asked |= {"MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW", "MCGPU_FLYABLE_LOW", "MCGPU_SWAP_BATCH"}  # read through an array of names
    assert len(asked) >= 30 and asked <= set(table), asked - set(table)
    assert {k for k, v in table.items() if v["scope"] in "KHT"} <= asked, "a registered engine knob nobody reads"
    # the Python side's variables are rows too
    for f in [ROOT / "4d-cbct-mc_amd" / "engine.py", ROOT / "tests" / "cases.py", ROOT / "tests" / "test_fast_rng.py"]:
        for name in re.findall(r'environ(?:\.get)?[\[(]\s*"(MCGPU_[A-Z0-9_]+)"', f.read_text()):
            assert name in table, (f.name, name)
    integration = (ROOT / "INTEGRATION.md").read_text()
    for name in table:
        assert f"`{name}`" in integration, f"{name} missing from INTEGRATION.md's knob table"
    gen = subprocess.run([sys.executable, str(ROOT / "tools" / "knob_table_md.py"), "--check"], timeout=120)
    assert gen.returncode == 0, "INTEGRATION.md section 6 is not the registry's table: run python tools/knob_table_md.py"
    exe = ROOT / "4d-cbct-mc_amd" / "MC-GPU_v1.3.x"
    listed = subprocess.run([str(exe), "--knobs"], capture_output=True, text=True, timeout=60)
    assert listed.returncode == 0 and all(name in listed.stdout for name in table)
    env = dict(os.environ, MCGPU_THRESH_COMPTN="32", MCGPU_SWAP_BATCH="40")
    r = subprocess.run([str(exe), str(tmp_path / "missing.in")], capture_output=True, text=True, timeout=60, env=env)
    warn = [l for l in r.stdout.split("\n") if "MCGPU_THRESH_COMPTN" in l]
    assert len(warn) == 1 and "ignored" in warn[0] and not re.search("(?i)error", warn[0]) and "MCGPU_SWAP_BATCH" not in r.stdout


def test_bench_builds_its_inputs_without_tests_and_oracle(tmp_path):
    """the input builder of the product's benchmark lives in the package (4d-cbct-mc_amd/workloads.py, tables under
    assets/); tests/ and oracle/ are checker-only imports of bench.py.  A tree without those two directories still imports bench.py,
    resolves the 22 material tables and the spectrum, and writes a workload's geometry + input file."""
    import subprocess
    import sys
    tree = tmp_path / "tree"
    tree.mkdir()
    import shutil
    for entry in ROOT.iterdir():  # bench.py, bench_legs/ and __graft_entry__.py as copies: their ROOT is the tree without tests/ and oracle/
        if entry.name in ("bench.py", "__graft_entry__.py"):
            shutil.copy(entry, tree / entry.name)
        elif entry.name == "bench_legs":
            shutil.copytree(entry, tree / entry.name, ignore=shutil.ignore_patterns("__pycache__"))
        elif entry.name in ("4d-cbct-mc_amd", "include"):
            (tree / entry.name).symlink_to(entry)
    code = ("import sys, bench\n"
            "from pathlib import Path\n"
            "pkg = bench.package()\n"
            "assert not any(Path(p).name in ('tests', 'oracle') for p in sys.path), sys.path\n"
            "m = pkg.workloads.material_files(); assert len(m) == 22 and all(f.is_file() and f.stat().st_size > 100000 for f in m)\n"
            "assert pkg.workloads.spectrum_file().is_file()\n"
            f"inp = bench.build_workload(Path({str(tmp_path / 'wl')!r}), 'catphan', 1000, 4, None, 32)\n"
            "assert inp.is_file() and (inp.parent / 'geometry.vox').is_file()\n"
            "print('ok', len(pkg.engine.knob_table()))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=tree, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MCGPU_TEST_CACHE=str(tmp_path / "cache"), PYTHONPATH=""))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
    text = (tmp_path / "wl" / "input.in").read_text()
    assert str(tmp_path / "cache" / "materials") in text and "geometry.vox" in text


def test_double_constants_of_the_fast64_kernel_are_what_their_comments_say():
    """csrc/track_common.inc spells its double constants as two 32-bit scalar words (`sconst(lo, hi) /* value */`: v_fma_f64 takes no
    64-bit literal and the compiler would otherwise hoist them into vector registers): the words are the IEEE-754 bits of the value."""
    import struct
    text = (ROOT / "4d-cbct-mc_amd" / "csrc" / "track_common.inc").read_text()
    found = re.findall(r"sconst\(0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u\) /\* ([^*]+?) \*/", text)
    assert len(found) >= 15
    for lo, hi, value in found:
        v = {"2 pi 2^-32": 6.28318530717958647693 * 2.0 ** -32}.get(value.strip())
        v = float(value) if v is None else v
        assert struct.unpack("<II", struct.pack("<d", v)) == (int(lo, 16), int(hi, 16)), value
    # ... and the polynomial coefficients of sincos_turn are the Taylor ones their trailing comments name (a 1/13! copied from a
    # minimax table once cost 7e-14 of accuracy at a quarter-turn border: found by the device known-answer test, kept out by this one)
    import math
    taylor = re.findall(r"/\* (-?[0-9.e+-]+) \*/\)?;\s+// (-?)1/(\d+)!", text)
    assert len(taylor) == 14
    for value, sign, k in taylor:
        assert float(value) == (-1.0 if sign else 1.0) / math.factorial(int(k)), (value, k)
