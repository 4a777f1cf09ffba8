"""Projection post-processing and MetaImage stacks against the reference's own Python recipe
(cbctmc/mc/projection.py:36-169) restated with numpy/scipy on the engine's ASCII files."""
from pathlib import Path

import numpy as np
import pytest
import scipy.ndimage as ndi

import cases


def _reference_read_raw(path, n_detector_pixels, half_fan_nx):
    """projection.py:36-51, verbatim in behaviour."""
    data = np.loadtxt(path, dtype=np.float64).astype(np.float32)
    data = data.reshape(*n_detector_pixels[::-1], 4)
    data = np.flip(data, axis=0)
    return data[:, :half_fan_nx] if half_fan_nx else data


def _reference_modes(stack4):
    """projection.py:118-133 for an array [n, Nz, Nx, 4]."""
    out = {}
    for mode in ("total", "unscattered", "scattered"):
        p = stack4.sum(axis=-1) if mode == "total" else stack4[..., 0] if mode == "unscattered" else stack4[..., 1:].sum(axis=-1)
        p = np.where(p == 0, p[p > 0.0].min(), p)
        out[mode] = p
    return out


def _random_tallies(rng, nz, nx, n):
    """Integer tallies with the texture of real ones: many zeros in the scatter classes, large primaries, tiny values."""
    img = np.zeros((n, 4, nz, nx), dtype=np.uint64)
    img[:, 0] = rng.integers(0, 5_000_000_000, (n, nz, nx))
    for k in (1, 2, 3):
        m = rng.uniform(size=(n, nz, nx)) < 0.3
        img[:, k][m] = rng.integers(1, 12_500_000, int(m.sum()))
    img[:, 0, 0, :7] = [0, 1, 2, 3, 49, 50, 51]  # around the 8th decimal
    return img


def test_finalize_matches_the_ascii_detour_bit_for_bit(engine, case_dir, tmp_path):
    rng = np.random.default_rng(11)
    with engine.create(case_dir("catphan64_ct"), device=-1) as ctx:
        nz, nx = ctx.detector_shape
        imgs = _random_tallies(rng, nz, nx, 3)
        for n_hist in (1, 22_500, 100_012_800):
            ref, got = [], []
            for p in range(3):
                f = tmp_path / f"proj_{n_hist}_{p}"
                ctx.write_projection(p, imgs[p], n_hist, file_name=str(f))
                ref.append(_reference_read_raw(f, (nx, nz), 128))
                got.append(ctx.finalize_host(imgs[p], n_hist, crop_nx=128))
            ref = np.stack(ref)
            got = np.stack(got)
            want_total, want_uns, want_sca = ref.sum(axis=-1), ref[..., 0], ref[..., 1:].sum(axis=-1)
            assert np.array_equal(got[:, 0].view(np.uint32), want_total.view(np.uint32))
            assert np.array_equal(got[:, 1].view(np.uint32), want_uns.view(np.uint32))
            assert np.array_equal(got[:, 2].view(np.uint32), want_sca.view(np.uint32))
        full = ctx.finalize_host(imgs[0], 22_500)
        assert full.shape == (3, nz, nx) and np.array_equal(full[:, :, :128], ctx.finalize_host(imgs[0], 22_500, crop_nx=128))


def test_stack_writer_and_zero_replacement(engine, tmp_path):
    rng = np.random.default_rng(5)
    planes = rng.uniform(0, 50, (5, 12, 20)).astype(np.float32)
    planes[rng.uniform(size=planes.shape) < 0.2] = 0.0
    w = engine.StackWriter(tmp_path / "s.mha", 20, 12, 5, spacing=(0.776, 0.776))
    for p in planes:
        w.append(p)
    fill = w.finish(replace_zeros=True)
    want = np.where(planes == 0, planes[planes > 0].min(), planes)
    assert fill == planes[planes > 0].min()
    assert np.array_equal(engine.stack_read(tmp_path / "s.mha"), want)
    header = (tmp_path / "s.mha").read_bytes().split(b"ElementDataFile = LOCAL\n")[0].decode()
    # what SimpleITK writes for GetImageFromArray + SetSpacing((sx, sy, 1)) + SetOrigin((-nx*sx/2, -ny*sy/2, 0)) (projection.py:155-164)
    assert "NDims = 3" in header and "DimSize = 20 12 5" in header and "ElementType = MET_FLOAT" in header
    assert "ElementSpacing = 0.776 0.776 1" in header
    off = [float(v) for v in header.split("Offset = ")[1].split("\n")[0].split()]
    assert off == [-20 * 0.776 / 2, -12 * 0.776 / 2, 0.0]  # the doubles Python computes in projection.py:158-163
    # a stack closed early or over-filled is an error, not a silently short file
    w = engine.StackWriter(tmp_path / "short.mha", 20, 12, 3)
    w.append(planes[0])
    with pytest.raises(engine.EngineError):
        w.finish()
    # all-zero stack: nothing to replace with
    w = engine.StackWriter(tmp_path / "z.mha", 4, 2, 1)
    w.append(np.zeros((2, 4), dtype=np.float32))
    assert np.isinf(w.finish(replace_zeros=True))
    assert not engine.stack_read(tmp_path / "z.mha").any()


def test_zero_replacement_of_slices_with_many_and_with_few_zeros(engine, tmp_path):
    """mha_finish (postprocess.cpp) rewrites slices that hold many zeros whole and patches a handful in place: planes whose size is
    no multiple of 64, half zeros, a slice with a handful of zeros (the position list), a slice without any, slices written by
    index in shuffled order, the smallest value arriving with the last slice, a caller who wants no replacement -- against numpy's
    `np.where(stack == 0, stack[stack > 0].min(), stack)` (projection.py:131-133)."""
    rng = np.random.default_rng(9)
    planes = rng.uniform(1, 50, (9, 33, 70)).astype(np.float32)
    planes[rng.uniform(size=planes.shape) < 0.5] = 0.0
    planes[3] = rng.uniform(1, 50, (33, 70)).astype(np.float32)
    planes[3].reshape(-1)[[0, 63, 64, 2309]] = 0.0      # few zeros, at word boundaries and at the very end
    planes[5] = rng.uniform(1, 50, (33, 70)).astype(np.float32)  # none
    planes[7].reshape(-1)[-70:] = 0.0                   # a whole last row (the tail word of the mask)
    want = np.where(planes == 0, planes[planes > 0].min(), planes)
    w = engine.StackWriter(tmp_path / "a.mha", 70, 33, 9)
    for p in planes:
        w.append(p)
    assert w.finish(replace_zeros=True) == planes[planes > 0].min()
    assert np.array_equal(engine.stack_read(tmp_path / "a.mha"), want)
    w = engine.StackWriter(tmp_path / "b.mha", 70, 33, 9)
    for k in rng.permutation(9):
        w.write_slice(int(k), planes[k])
    w.finish(replace_zeros=True)
    assert np.array_equal(engine.stack_read(tmp_path / "b.mha"), want)
    assert (tmp_path / "a.mha").read_bytes() == (tmp_path / "b.mha").read_bytes()
    # a caller who wants no replacement keeps its zeros
    w = engine.StackWriter(tmp_path / "c.mha", 70, 33, 9)
    for p in planes:
        w.append(p)
    w.finish(replace_zeros=False)
    assert np.array_equal(engine.stack_read(tmp_path / "c.mha"), planes)
    # the smallest value arrives LAST
    late = planes.copy()
    late[8].reshape(-1)[5] = 0.25
    w = engine.StackWriter(tmp_path / "d.mha", 70, 33, 9)
    for p in late:
        w.append(p)
    assert w.finish(replace_zeros=True) == np.float32(0.25)
    assert np.array_equal(engine.stack_read(tmp_path / "d.mha"), np.where(late == 0, np.float32(0.25), late))


def test_air_normalisation_matches_scipy_recipe(engine, tmp_path):
    """normalize_projections (projection.py:96-115): log(gaussian_filter(air, sigma) / projections), float32 throughout."""
    rng = np.random.default_rng(9)
    ny, nx = 96, 128
    air = (200.0 + 20.0 * rng.standard_normal((ny, nx))).astype(np.float32)
    proj = rng.uniform(0.5, 180.0, (4, ny, nx)).astype(np.float32)
    for name, arr in (("air.mha", air[None]), ("total.mha", proj)):
        w = engine.StackWriter(tmp_path / name, nx, ny, arr.shape[0])
        for p in arr:
            w.append(p)
        w.finish(replace_zeros=False)
    for sigma in ((10, 10), (3, 7), None):
        engine.normalize_stack(tmp_path / "total.mha", tmp_path / "air.mha", tmp_path / "norm.mha", sigma=sigma)
        air_f = ndi.gaussian_filter(air, sigma=sigma) if sigma else air
        want = np.log(air_f / proj)
        got = engine.stack_read(tmp_path / "norm.mha")
        # tolerance: float32 log of two libraries (numpy SIMD vs libm), <= 2 ulp; the filter itself is bit-exact (below)
        assert np.allclose(got, want, rtol=3e-7, atol=3e-7), sigma
    # the gaussian filter alone, through log(filtered / 1): compare exp() of it loosely and the filter exactly via a unit stack
    ones = np.ones((1, ny, nx), dtype=np.float32)
    w = engine.StackWriter(tmp_path / "ones.mha", nx, ny, 1)
    w.append(ones[0])
    w.finish(replace_zeros=False)
    engine.normalize_stack(tmp_path / "ones.mha", tmp_path / "air.mha", tmp_path / "f.mha", sigma=(10, 10))
    got = engine.stack_read(tmp_path / "f.mha")[0]
    assert np.allclose(got, np.log(ndi.gaussian_filter(air, sigma=(10, 10))), rtol=3e-7)


def test_postprocess_pipeline_equals_reference_recipe(engine, case_dir, tmp_path):
    """Stacks written by the engine == the reference's postprocess_simulation recipe applied to the engine's ASCII files."""
    rng = np.random.default_rng(21)
    with engine.create(case_dir("catphan64_ct"), device=-1) as ctx:
        nz, nx = ctx.detector_shape
        imgs = _random_tallies(rng, nz, nx, 4)
        n_hist = 60_000
        ref4 = []
        writers = {m: engine.StackWriter(tmp_path / f"projections_{m}.mha", 100, nz, 4) for m in ("total", "unscattered", "scattered")}
        for p in range(4):
            f = tmp_path / ctx.projection_file_name(p).split("/")[-1]
            ctx.write_projection(p, imgs[p], n_hist, file_name=str(f))
            ref4.append(_reference_read_raw(f, (nx, nz), 100))
            planes = ctx.finalize_host(imgs[p], n_hist, crop_nx=100)
            for k, m in enumerate(("total", "unscattered", "scattered")):
                writers[m].append(planes[k])
        want = _reference_modes(np.stack(ref4))
        for m, w in writers.items():
            w.finish(replace_zeros=True)
            got = engine.stack_read(tmp_path / f"projections_{m}.mha")
            assert np.array_equal(got.view(np.uint32), want[m].view(np.uint32)), m


@pytest.mark.gpu
def test_device_finalize_and_scan_pipeline(engine, case_dir, tmp_path):
    """finalize kernel == host finalize; run_scan's files == projection-by-projection results."""
    import torch
    n_hist = 300_000
    (tmp_path / "air_out").mkdir()
    with engine.create(case_dir("air"), device=0) as actx:  # the air scan of the reference flow (simulation.py:429-470)
        actx.run_scan(histories=2_000_000, crop_nx=128, output_folder=tmp_path / "air_out")
    air_stack = tmp_path / "air_out" / "projections_total.mha"
    assert engine.stack_read(air_stack).shape[0] == 1
    with engine.create(case_dir("catphan64_ct"), device=0) as ctx:
        nz, nx = ctx.detector_shape
        # device finalize vs host finalize on a real tally, and the fused clear
        img, _, done = ctx.run_projection(1, n_hist, mode="fast", seed=ctx.geti("seed"))
        dev = torch.from_numpy(img.astype(np.int64)).cuda()
        planes = torch.zeros((3, nz, 128), dtype=torch.float32, device="cuda")
        ctx.finalize_device(dev.data_ptr(), done, planes.data_ptr(), crop_nx=128, clear=True)
        torch.cuda.synchronize()
        assert np.array_equal(planes.cpu().numpy().view(np.uint32), ctx.finalize_host(img, done, crop_nx=128).view(np.uint32))
        assert int(dev.abs().sum().item()) == 0
        # the scan pipeline: stacks + ASCII files + air normalisation
        out = tmp_path / "scan"
        out.mkdir()
        rep = ctx.run_scan(mode="fast", histories=n_hist, crop_nx=128, write_ascii=True, output_folder=out, air_stack=air_stack, air_sigma=(10, 10))
        assert rep["projections"] == 4 and rep["histories_per_projection"] == n_hist
        assert 0.0 < rep["kernel_ms_min"] <= rep["kernel_ms_max"] and rep["kernel_ms_min"] * 4 <= rep["seconds_kernels"] * 1e3 * (1 + 1e-6) <= rep["kernel_ms_max"] * 4 * (1 + 2e-6)
        per_proj = [ctx.run_projection(p, n_hist, mode="fast", seed=ctx.geti("seed"))[0] for p in range(4)]
        planes = np.stack([ctx.finalize_host(i, n_hist, crop_nx=128) for i in per_proj])  # [4, 3, nz, 128]
        for k, m in enumerate(("total", "unscattered", "scattered")):
            want = planes[:, k]
            want = np.where(want == 0, want[want > 0].min(), want)
            got = engine.stack_read(out / f"projections_{m}.mha")
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), m
        tot = engine.stack_read(out / "projections_total.mha")
        air = ndi.gaussian_filter(engine.stack_read(air_stack)[0], sigma=(10, 10))
        assert np.allclose(engine.stack_read(out / "projections_total_normalized.mha"), np.log(air / tot), rtol=3e-7, atol=3e-7)
        # the ASCII files of the scan are the ones write_projection gives for the same tallies
        for p in range(4):
            name = ctx.projection_file_name(p)
            ref_file = tmp_path / f"ref_{p}"
            ctx.write_projection(p, per_proj[p], n_hist, file_name=str(ref_file))
            data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]  # footer: the speed line is optional
            assert data(name) == data(ref_file)


@pytest.mark.gpu
@pytest.mark.parametrize("writers", [1, 2])
def test_scan_ascii_files_with_few_formatter_slots(engine, case_dir, tmp_path, monkeypatch, writers):
    """MCGPU_ASCII_WRITERS=1: every projection is formatted into the ONE slot its predecessor used -- the projection loop
    must wait until that predecessor has been handed to the slot's worker and written (scan.cpp: enqueue_reduce).  The files
    of the scan equal the host formatter's for the same tallies."""
    n_hist = 200_000
    monkeypatch.setenv("MCGPU_ASCII_WRITERS", str(writers))
    with engine.create(case_dir("catphan64_ct", n_projections=8, angle_between_projections=45.0), device=0) as ctx:
        out = tmp_path / "scan"
        out.mkdir()
        rep = ctx.run_scan(mode="fast", histories=n_hist, crop_nx=128, write_ascii=True, write_stacks=False, output_folder=out)
        assert rep["projections"] == 8
        data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]
        for p in range(8):
            img = ctx.run_projection(p, n_hist, mode="fast", seed=ctx.geti("seed"))[0]
            ref_file = tmp_path / f"ref_{p}"
            ctx.write_projection(p, img, n_hist, file_name=str(ref_file))
            assert data(ctx.projection_file_name(p)) == data(ref_file), p


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["1", "0"])
def test_scan_sharded_over_contexts_equals_single_context(engine, case_dir, tmp_path, monkeypatch, policy):
    """mcgpu_run_scan_multi: three contexts (here on one device) shard the histories; every projection's tallies are summed on
    its owner through the tally exchange (exchange.cpp: copy-engine pushes, one fused add), finalized and written there -- the
    owner rotates over the devices (policy 1) or is always the first one (policy 0, the reference's root).  Stacks and ASCII
    files equal the single-context scan bit for bit."""
    monkeypatch.setenv("MCGPU_EXCHANGE_POLICY", policy)
    n_hist = 500_000
    outs = []
    for k, n_peers in enumerate((0, 2)):
        out = tmp_path / f"scan{k}"
        out.mkdir()
        ctxs = [engine.create(case_dir("catphan64_ct"), device=0) for _ in range(1 + n_peers)]
        try:
            rep = ctxs[0].run_scan(mode="fast", histories=n_hist, crop_nx=128, write_ascii=True, output_folder=out, peers=ctxs[1:])
            assert rep["projections"] == 4 and rep["histories_per_projection"] == n_hist
            # fastest / slowest projection of the scan (mcgpu_scan_report.kernel_ms_min / _max) bracket the mean kernel time
            assert 0.0 < rep["kernel_ms_min"] <= rep["seconds_kernels"] * 1e3 / 4 * (1 + 1e-6) and rep["seconds_kernels"] * 1e3 / 4 <= rep["kernel_ms_max"] * (1 + 1e-6)
            ascii_files = [Path(ctxs[0].projection_file_name(p)).read_bytes() for p in range(4)]
        finally:
            for c in ctxs:
                c.close()
        outs.append({m: engine.stack_read(out / f"projections_{m}.mha") for m in ("total", "unscattered", "scattered")})
        outs[-1]["ascii"] = [[l for l in t.split(b"\n") if not l.startswith(b"#")] for t in ascii_files]  # data lines (the footer times differ)
    for m in ("total", "unscattered", "scattered"):
        assert np.array_equal(outs[0][m], outs[1][m]), m
    assert outs[0]["ascii"] == outs[1]["ascii"] and len(outs[0]["ascii"][0]) > 1000
    # COMPAT mode shards RANECU batches: same statement
    res = []
    for n_peers in (0, 1):
        out = tmp_path / f"compat{n_peers}"
        out.mkdir()
        ctxs = [engine.create(case_dir("catphan64_ct"), device=0) for _ in range(1 + n_peers)]
        try:
            ctxs[0].run_scan(mode="compat", histories=19200 * 5, crop_nx=0, output_folder=out, peers=ctxs[1:])
        finally:
            for c in ctxs:
                c.close()
        res.append(engine.stack_read(out / "projections_total.mha"))
    assert np.array_equal(res[0], res[1])


@pytest.mark.gpu
def test_scan_sharded_by_projection_equals_single_context(engine, case_dir, tmp_path):
    """mcgpu_run_scan_multi with MCGPU_SHARD_PROJECTIONS (SURVEY 8e's fallback): three contexts, each simulating whole projections
    on its own pipeline thread, shared stacks filled by slice index; and the building block it is made of -- one plain
    mcgpu_run_scan per context with projection_stride / projection_phase into caller-owned stacks.  Stacks equal the
    single-context scan bit for bit in both personalities."""
    for mode, hist in (("fast", 400_000), ("compat", 19200 * 5)):
        ref_dir, out = tmp_path / f"{mode}_one", tmp_path / f"{mode}_three"
        ref_dir.mkdir(); out.mkdir()
        with engine.create(case_dir("catphan64_ct"), device=0) as c:
            c.run_scan(mode=mode, histories=hist, crop_nx=128, output_folder=ref_dir)
        ctxs = [engine.create(case_dir("catphan64_ct"), device=0) for _ in range(3)]
        try:
            rep = ctxs[0].run_scan(mode=mode, histories=hist, crop_nx=128, output_folder=out, peers=ctxs[1:], shard="projections")
            assert rep["projections"] == 4
        finally:
            for c in ctxs:
                c.close()
        for m in ("total", "unscattered", "scattered"):
            assert np.array_equal(engine.stack_read(ref_dir / f"projections_{m}.mha"), engine.stack_read(out / f"projections_{m}.mha")), (mode, m)


@pytest.mark.gpu
def test_microbench_reports_plausible_ceilings(engine, case_dir):
    """mcgpu_microbench: the ceilings bench.py quotes are measured, positive and in the range the part can deliver (an MI355X issues
    between 0.4 and 1.3 vector wave-instructions per ns and SIMD, and a few 1e10 scattered atomics per second)."""
    with engine.create(case_dir("air"), device=0) as ctx:
        v = ctx.microbench("valu_issue")
        a = ctx.microbench("atomic_rate")
    assert len(v) == 3 and all(0.4 < x < 1.3 for x in v), v
    assert v[1] >= 0.95 * v[0]  # half the lanes idle never issues slower
    assert 5e9 < a < 1e11, a


@pytest.mark.gpu
def test_device_formatted_projection_file_is_byte_identical(engine, case_dir, tmp_path):
    """The data lines formatted on the device ("%.8lf" by exact integer arithmetic, ascii_device.hip) against the host writer
    (itself byte-identical to the reference's report_image on the data lines): random tallies over many magnitudes, zeros, the
    largest words, and values that land exactly on a rounding tie."""
    import torch
    rng = np.random.default_rng(11)
    with engine.create(case_dir("catphan64_ct"), device=0) as ctx:
        nz, nx = ctx.detector_shape
        for k, n_hist in enumerate((1_000_000, 3, 2 ** 20, 123_456_789)):
            norm = 0.01 * (10.0 / ctx.getf("pixel_size_x_mm")) * (10.0 / ctx.getf("pixel_size_z_mm")) / n_hist
            top = min(int(np.log2(9.0e10 / norm)), 62)  # values up to 9e10 eV/cm^2 per history: 11 integer digits (physical ones have <= 6)
            img = np.zeros((4, nz, nx), dtype=np.uint64)
            mag = rng.integers(0, top, size=img.shape)
            img[:] = (rng.random(img.shape) * (2.0 ** mag)).astype(np.uint64)
            img[rng.random(img.shape) < 0.3] = 0
            img[0, 0, :6] = [0, 1, 2, 5, 2 ** (top - 1), 2 ** top - 1]
            if n_hist == 2 ** 20:
                # many products NORM * e * 1e8 with short binary fractions (ties and near-ties of the 8th digit among them)
                img[1, 1, :] = np.arange(nx, dtype=np.uint64) * 5 + 1
                img[2, 2, :] = (np.arange(nx, dtype=np.uint64) + 1) * 2 ** 19
            dev = torch.from_numpy(img.view(np.int64)).cuda()
            want = tmp_path / f"host_{k}"
            got = tmp_path / f"device_{k}"
            ctx.write_projection(1, img, n_hist, 2.5, file_name=str(want))
            ctx.write_projection_device(1, dev.data_ptr(), n_hist, 2.5, file_name=str(got), slot=k % 3)
            a, b = want.read_bytes(), got.read_bytes()
            assert len(a) == len(b) and a == b, (k, len(a), len(b))
        # 12 integer digits: outside the formatter's buffers -- refused, nothing is written
        img[:] = 0
        n_hist = 3
        img[3, 5, 7] = int(2.0e11 / (0.01 * (10.0 / ctx.getf("pixel_size_x_mm")) * (10.0 / ctx.getf("pixel_size_z_mm")) / n_hist))
        dev = torch.from_numpy(img.view(np.int64)).cuda()
        with pytest.raises(engine.EngineError) as e:
            ctx.write_projection_device(1, dev.data_ptr(), n_hist, 2.5, file_name=str(tmp_path / "refused"), slot=0)
        assert e.value.code == -3 and not (tmp_path / "refused").exists()
