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
    inp = cases.build_case("catphan64_ct", tmp_path, n_histories=19200 * 5)  # >= 95000: below that the value is a time budget
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


def test_executable_skips_projections_outside_the_angular_roi(engine, tmp_path):
    """MC-GPU_v1.3.cu:670-677: projections outside "ANGLES OF INTEREST" are not simulated, write no file, and do not
    advance the seed (:869 sits behind the `continue`)."""
    inp = cases.build_case("catphan64_ct", tmp_path, n_histories=19200 * 5)
    text = inp.read_text()
    assert "0.0 5000.0  # ANGLES OF INTEREST" in text
    inp.write_text(text.replace("0.0 5000.0  # ANGLES OF INTEREST", "300.0 500.0  # ANGLES OF INTEREST"))
    res = subprocess.run([str(engine.EXE_PATH), str(inp), "--mode", "compat", "--stacks", "--crop", "0"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and not re.search("(?i)error", res.stdout), res.stdout[-2000:]
    assert [int(a) for a in re.findall(r"Skipping projection #(\d+) of 4", res.stdout)] == [1, 4]
    assert [int(a) for a in re.findall(r"Simulating Projection (\d+) of 4", res.stdout)] == [2, 3]
    files = sorted(f.name for f in tmp_path.iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert files == ["projection_360.000000deg", "projection_450.000000deg"]
    assert engine.stack_read(tmp_path / "projections_total.mha").shape[0] == 2
    with engine.create(inp, device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        nz, nx = ctx.detector_shape
        batches, hpt, total = ctx.reference_shape()
        det = ctx.host_table("detector_data", "<f4")
        norm = 0.01 * float(det[19]) * float(det[20]) / total
        seed = 42
        for p, name in zip((1, 2), files):
            img, _ = T.track(p, seed, 0, batches, hpt, ol.MATH_PORTABLE, n_threads=4)
            want = np.flip((img.reshape(4, nz, nx).astype(np.float64) * norm).transpose(1, 2, 0), axis=0)
            assert np.allclose(_read_like_reference(tmp_path / name, nz, nx, np.float64), want, rtol=0, atol=0.51e-8)
            seed = engine.advance_seed(1, total, seed)


def test_executable_reports_errors_like_the_reference(engine, tmp_path):
    res = subprocess.run([str(engine.EXE_PATH), str(tmp_path / "missing.in")], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and re.search("(?i)error", res.stdout)
    res = subprocess.run([str(engine.EXE_PATH)], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "ERROR" in res.stdout


def test_mcsimulation_mirror_runs_air_and_object_scans(engine, tmp_path):
    """MCSimulation.run_simulation flow (sim.py:370-427): air scan + object scan + post-processing -> RTK-ready stacks;
    log(air/object) is a sane attenuation image."""
    mats, spc = cases.material_files(), cases.spectrum_file()
    kw = dict(n_projections=1, n_detector_pixels=(231, 96), detector_size=(717.024, 297.984))
    obj = cases.simulation.MCSimulation(cases.geometry.MCBoxGeometry(shape=(20, 20, 20), image_spacing=(10.0, 10.0, 10.0)), mats, spc,
                                        n_histories=4_000_000, **kw)
    rep = obj.run_simulation(tmp_path / "obj", engine, mode="fast", run_air_simulation=True, air_n_histories=4_000_000, clean=False,
                             air_projection_denoise_kernel_size=None)
    assert rep["projections"] == 1 and rep["histories_per_projection"] == 4_000_000
    assert obj.run_simulation(tmp_path / "obj", engine) is None, "an output folder with stacks counts as simulated (sim.py:389-395)"
    nz, nx = 96, 231
    eng = engine
    # the stacks hold what the reference's Python reads from the ASCII files (kept because clean=False)
    (name_o,) = [f for f in (tmp_path / "obj").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name)]
    assert not [f for f in (tmp_path / "obj" / "air").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name)], \
        "the air scan runs with clean=True like the reference's (sim.py:85-87)"
    o = _read_like_reference(name_o, nz, nx).sum(axis=-1)
    o_stack = eng.stack_read(tmp_path / "obj" / "projections_total.mha")[0]
    a_stack = eng.stack_read(tmp_path / "obj" / "air" / "projections_total.mha")[0]
    a = a_stack
    assert np.array_equal(o_stack, np.where(o == 0, o[o > 0].min(), o))
    norm = eng.stack_read(tmp_path / "obj" / "projections_total_normalized.mha")[0]
    assert np.allclose(norm, np.log(a_stack / o_stack), rtol=3e-7, atol=3e-7)
    # the same two scans on the CPU oracle (fewer histories), read through the same normalisation
    def oracle_total(sim_dir, n_batches=10000, hpt=150):
        with engine.create(sim_dir / "input.in", device=-1) as ctx:
            T = parity.tables_from_context(ctx)
            det = ctx.host_table("detector_data", "<f4")
            img, _ = T.track(0, 42, 0, n_batches, hpt, ol.MATH_LIBM, n_threads=16)
            norm = 0.01 * float(det[19]) * float(det[20]) / (n_batches * hpt)
            return np.flip(img.reshape(4, nz, nx).sum(axis=0).astype(np.float64) * norm, axis=0)
    a_ref, o_ref = oracle_total(tmp_path / "obj" / "air"), oracle_total(tmp_path / "obj")
    # masks from the (22x better sampled) GPU images, block-averaged 8x7 so that selection is not noise-driven
    blk = lambda im: im[: nz // 8 * 8, : nx // 7 * 7].reshape(nz // 8, 8, nx // 7, 7).mean(axis=(1, 3))
    ab, ob, abr, obr = blk(a), blk(o), blk(a_ref), blk(o_ref)
    lit = ab > 0.5 * ab.max()  # illuminated half-fan region
    shadow = lit & (ob < 0.1 * ab)
    assert shadow.sum() >= 20 and lit.sum() >= 60
    mu_gpu = np.log(ab[shadow].mean() / ob[shadow].mean())
    mu_ref = np.log(abr[shadow].mean() / obr[shadow].mean())
    assert 2.3 < mu_ref < 5.0, mu_ref  # ~20 cm of water incl. scatter build-up
    assert abs(mu_gpu - mu_ref) < 0.03 * mu_ref, (mu_gpu, mu_ref)
    assert abs(ab[lit].mean() / abr[lit].mean() - 1.0) < 0.01


def test_executable_sharded_over_devices_equals_single_device(engine, tmp_path):
    """`--gpus N` path of the drop-in executable (one host thread per device, integer sum of the tallies, the reference's
    MPI_Reduce): run as two shards on device 0, it must reproduce the single-device files bit for bit -- ASCII and stacks."""
    a = cases.build_case("catphan64_ct", tmp_path / "one", n_histories=400_000)
    b = cases.build_case("catphan64_ct", tmp_path / "two", n_histories=400_000)
    r1 = subprocess.run([str(engine.EXE_PATH), str(a), "--stacks", "--crop", "128"], capture_output=True, text=True, timeout=600)
    r2 = subprocess.run([str(engine.EXE_PATH), str(b), "--devices", "0,0", "--stacks", "--crop", "128"], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stdout[-2000:] + r2.stdout[-2000:]
    assert not re.search("(?i)error", r1.stdout + r2.stdout)
    names = sorted(f.name for f in (tmp_path / "one").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert len(names) == 4
    data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]  # footer: the speed line is optional
    for n in names:
        assert data(tmp_path / "one" / n) == data(tmp_path / "two" / n), n
    for m in ("total", "unscattered", "scattered"):
        s1, s2 = engine.stack_read(tmp_path / "one" / f"projections_{m}.mha"), engine.stack_read(tmp_path / "two" / f"projections_{m}.mha")
        assert s1.shape == (4, 96, 128) and np.array_equal(s1, s2), m


def test_executable_eight_ranks_on_one_device(engine, tmp_path):
    """BASELINE configs 3 and 5 ask for 8 GPUs; the box has one.  The executable's `--gpus 8` path on eight contexts of device 0 -- the
    tally exchange with a rotating owner and seven pushes per projection, and projection sharding with eight uneven shares of eleven
    projections -- writes the single-device files byte for byte.  (Eight rank PROCESSES cannot share the box's GPU: the pool admits
    six processes on a card; bench.py's multi-process path is rehearsed with 2-6 ranks, profiles/r06*_ranks_sharing_one_gpu.json.)"""
    kw = dict(n_histories=400_000, n_projections=11, angle_between_projections=33.0)
    dirs = {k: cases.build_case("catphan64_ct", tmp_path / k, **kw) for k in ("one", "exchange8", "projections8")}
    common = ["--stacks", "--crop", "128"]
    eight = ["--devices", "0,0,0,0,0,0,0,0"]
    runs = {"one": subprocess.run([str(engine.EXE_PATH), str(dirs["one"])] + common, capture_output=True, text=True, timeout=900),
            "exchange8": subprocess.run([str(engine.EXE_PATH), str(dirs["exchange8"])] + eight + common, capture_output=True, text=True, timeout=900),
            "projections8": subprocess.run([str(engine.EXE_PATH), str(dirs["projections8"])] + eight + ["--shard", "projections"] + common, capture_output=True, text=True, timeout=900)}
    for k, r in runs.items():
        assert r.returncode == 0, (k, r.stdout[-2000:] + r.stderr[-2000:])
        assert not re.search("(?i)error", r.stdout), (k, r.stdout[-1500:])
    assert "not available" not in runs["exchange8"].stdout  # the exchange itself ran, no fallback
    names = sorted(f.name for f in (tmp_path / "one").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert len(names) == 11
    data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]
    for k in ("exchange8", "projections8"):
        for n in names:
            assert data(tmp_path / "one" / n) == data(tmp_path / k / n), (k, n)
        for m in ("total", "unscattered", "scattered"):
            assert np.array_equal(engine.stack_read(tmp_path / "one" / f"projections_{m}.mha"), engine.stack_read(tmp_path / k / f"projections_{m}.mha")), (k, m)


@pytest.mark.parametrize("mode", ["fast", "compat"])
def test_executable_sharded_by_projection_equals_single_device(engine, tmp_path, mode):
    """`--shard projections` (SURVEY 8e's fallback: every device simulates whole projections, nothing crosses between devices):
    three "devices" over eight projections, the first of them outside the angular region of interest -- seven simulated, so the
    shares are uneven and the skipped projection shifts the ownership, and in COMPAT mode the seed must still move on once per simulated projection
    whoever simulates it (MC-GPU_v1.3.cu:869).  Files and stacks must equal the single-device run's byte for byte."""
    kw = dict(n_histories=120_000, n_projections=8, angle_between_projections=50.0)
    a = cases.build_case("catphan64_ct", tmp_path / "one", **kw)
    b = cases.build_case("catphan64_ct", tmp_path / "three", **kw)
    for inp in (a, b):  # the trajectory starts at 270 degrees: an angular ROI from 300 degrees drops projection #1
        text = inp.read_text()
        assert "0.0 5000.0  # ANGLES OF INTEREST" in text
        inp.write_text(text.replace("0.0 5000.0  # ANGLES OF INTEREST", "300.0 5000.0  # ANGLES OF INTEREST"))
    common = ["--stacks", "--crop", "128", "--mode", mode]
    r1 = subprocess.run([str(engine.EXE_PATH), str(a)] + common, capture_output=True, text=True, timeout=600)
    r3 = subprocess.run([str(engine.EXE_PATH), str(b), "--devices", "0,0,0", "--shard", "projections"] + common, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and r3.returncode == 0, r1.stdout[-2000:] + r3.stdout[-2000:]
    assert not re.search("(?i)error", r1.stdout + r3.stdout)
    assert sorted(int(v) for v in re.findall(r"<< Simulating Projection (\d+) of 8 >>", r3.stdout)) == [2, 3, 4, 5, 6, 7, 8]
    assert [int(v) for v in re.findall(r"Skipping projection #(\d+) of 8", r3.stdout)] == [1]
    names = sorted(f.name for f in (tmp_path / "one").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert len(names) == 7
    data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]
    for n in names:
        assert data(tmp_path / "one" / n) == data(tmp_path / "three" / n), n
    for m in ("total", "unscattered", "scattered"):
        s1, s3 = engine.stack_read(tmp_path / "one" / f"projections_{m}.mha"), engine.stack_read(tmp_path / "three" / f"projections_{m}.mha")
        assert s1.shape == (7, 96, 128) and np.array_equal(s1, s3), m


def test_executable_falls_back_to_projection_sharding_without_an_exchange(engine, tmp_path):
    """A node whose devices cannot reach each other (no peer access, no copy-engine path: here requested through the test hook
    MCGPU_EXCHANGE_FAIL_PROBE) must not stop a `--gpus N` run: the set-up phase reports it before anything was simulated and the
    scan is sharded by projection instead -- same files as the single-device run, one line in the log that says so, and the word
    "error" nowhere (cbctmc/mc/simulation.py:204 greps for it)."""
    import os
    a = cases.build_case("catphan64_ct", tmp_path / "one", n_histories=300_000)
    b = cases.build_case("catphan64_ct", tmp_path / "two", n_histories=300_000)
    r1 = subprocess.run([str(engine.EXE_PATH), str(a), "--stacks", "--crop", "128"], capture_output=True, text=True, timeout=600)
    r2 = subprocess.run([str(engine.EXE_PATH), str(b), "--devices", "0,0", "--stacks", "--crop", "128"], capture_output=True, text=True, timeout=600,
                        env=dict(os.environ, MCGPU_EXCHANGE_FAIL_PROBE="1"))
    assert r1.returncode == 0 and r2.returncode == 0, r1.stdout[-2000:] + r2.stdout[-2000:]
    # AUTO: exchange (probe fails) -> projection sharding; the RCCL route is taken only when asked for (it has never run
    # over more than one rank, and without peer access it would stage every tally through host memory)
    assert "Tally exchange between the devices is not available: every device simulates whole projections instead" in r2.stdout
    assert "RCCL reduction (uint64, sum)" not in r2.stdout and "listed twice" not in r2.stdout
    assert not re.search("(?i)error", r1.stdout) and not re.search("(?i)error", r2.stdout), r2.stdout[-1500:]
    for m in ("total", "unscattered", "scattered"):
        assert np.array_equal(engine.stack_read(tmp_path / "one" / f"projections_{m}.mha"), engine.stack_read(tmp_path / "two" / f"projections_{m}.mha")), m


def test_executable_rccl_reduction_route(engine, tmp_path):
    """`--reduce rccl`: the per-projection sum of the device tallies as ONE ncclReduce(uint64, sum, root = owner) on a stream of its own
    beside the next projection's kernel (north_star's collective; the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019).  A one-GPU box
    cannot cross a link, but a communicator of one rank runs everything else for real: the library opened with dlopen,
    ncclCommInitAll, the grouped ncclReduce, the events between tracking and reduce stream, the double-buffered tally.  Files and
    stacks equal the plain run's byte for byte; with the device listed twice RCCL refuses (one rank per GPU) and the scan says so
    and shards by projection -- same bytes again."""
    import os
    kw = dict(n_histories=300_000, n_projections=5, angle_between_projections=72.0)
    dirs = {k: cases.build_case("catphan64_ct", tmp_path / k, **kw) for k in ("plain", "rccl", "rccl2")}
    common = ["--stacks", "--crop", "128"]
    r0 = subprocess.run([str(engine.EXE_PATH), str(dirs["plain"])] + common, capture_output=True, text=True, timeout=600)
    r1 = subprocess.run([str(engine.EXE_PATH), str(dirs["rccl"]), "--reduce", "rccl"] + common, capture_output=True, text=True, timeout=600)
    r2 = subprocess.run([str(engine.EXE_PATH), str(dirs["rccl2"]), "--devices", "0,0", "--reduce", "rccl"] + common, capture_output=True, text=True, timeout=600)
    for r in (r0, r1, r2):
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert not re.search("(?i)error", r.stdout), r.stdout[-1500:]
    assert "Detector tallies summed with one RCCL reduction (uint64, sum) per projection" in r1.stdout
    assert "The RCCL reduction that was asked for is not available: every device simulates whole projections instead" in r2.stdout and "listed twice" in r2.stdout
    names = sorted(f.name for f in (tmp_path / "plain").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert len(names) == 5
    data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]
    for k in ("rccl", "rccl2"):
        for n in names:
            assert data(tmp_path / "plain" / n) == data(tmp_path / k / n), (k, n)
        for m in ("total", "unscattered", "scattered"):
            assert np.array_equal(engine.stack_read(tmp_path / "plain" / f"projections_{m}.mha"), engine.stack_read(tmp_path / k / f"projections_{m}.mha")), (k, m)


def test_reference_command_line_through_the_mpirun_shim(engine, tmp_path):
    """What `MCSimulation.run_simulation` executes inside the container (cbctmc/mc/simulation.py:187-198): `mpirun --tag-output
    -v -n <gpus> MC-GPU_v1.3.x <input>`, its stdout scanned for progress lines and for the word "error", its projection files
    parsed afterwards.  Here the shim of the ROCm image (docker/mpirun) runs the real executable on the box's GPU; the files
    equal those of the executable called directly."""
    from pathlib import Path
    shim = Path(cases.ROOT) / "docker" / "mpirun"
    a = cases.build_case("catphan64_ct", tmp_path / "shim", n_histories=300_000)
    b = cases.build_case("catphan64_ct", tmp_path / "direct", n_histories=300_000)
    r1 = subprocess.run(["sh", str(shim), "--tag-output", "-v", "-n", "1", str(engine.EXE_PATH), str(a)], capture_output=True, text=True, timeout=600)
    r2 = subprocess.run([str(engine.EXE_PATH), str(b)], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    assert not re.search("(?i)error", r1.stdout)
    found = re.findall(r"Simulating Projection (\d{1,4}) of (\d{1,4})", r1.stdout)
    assert [int(i) for i, _ in found] == [1, 2, 3, 4]
    names = sorted(f.name for f in (tmp_path / "shim").iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name))
    assert len(names) == 4
    data = lambda f: [l for l in open(f).read().rstrip("\n").split("\n") if not l.startswith("#")]
    for n in names:
        assert data(tmp_path / "shim" / n) == data(tmp_path / "direct" / n), n
        assert _read_like_reference(tmp_path / "shim" / n, 96, 231).sum() > 0
    # more ranks than the box has GPUs: the reference's mpirun would fail to bind its ranks; the shim's executable says so and
    # the caller's scan for "error" catches it
    r3 = subprocess.run(["sh", str(shim), "-n", "3", str(engine.EXE_PATH), str(a)], capture_output=True, text=True, timeout=600)
    assert r3.returncode != 0 and re.search("(?i)error", r3.stdout)


def test_executable_time_limited_run(engine, tmp_path):
    """An input "number of histories" below 95000 is a time budget in seconds per projection (MC-GPU_v1.3.cu:650-655)."""
    inp = cases.build_case("water", tmp_path, n_histories=1)
    res = subprocess.run([str(engine.EXE_PATH), str(inp)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and not re.search("(?i)error", res.stdout), res.stdout[-2000:]
    m = re.search(r"Time-limited run: 1 s per projection at ([0-9.e+]+) x-rays/s -> (\d+) histories", res.stdout)
    assert m, res.stdout[-2000:]
    rate, n = float(m.group(1)), int(m.group(2))
    assert 0.5 * rate < n < 1.5 * rate and n > 1_000_000
    (name,) = [f for f in tmp_path.iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name)]
    assert f"Simulated x rays:    {n}" in name.read_text()


def test_executable_time_limited_run_sharded_by_projection_calibrates_once(engine, tmp_path):
    """A time budget (history count below 95000 = seconds per projection) with `--shard projections`: ONE calibration on the first
    device serves every shard -- each device calibrating by itself would simulate its own history count and the files would depend
    on who wrote them.  Every projection file reports the same number of simulated x rays, the one the log names."""
    inp = cases.build_case("water", tmp_path, n_histories=1, n_projections=4, angle_between_projections=90.0)
    res = subprocess.run([str(engine.EXE_PATH), str(inp), "--devices", "0,0", "--shard", "projections"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and not re.search("(?i)error", res.stdout), res.stdout[-2000:]
    lines = re.findall(r"Time-limited run: 1 s per projection at [0-9.e+]+ x-rays/s per device -> (\d+) histories per projection", res.stdout)
    assert len(lines) == 1, res.stdout[-2000:]
    n = int(lines[0])
    names = [f for f in tmp_path.iterdir() if cases.simulation.PROJECTION_FILE_PATTERN.match(f.name)]
    assert len(names) == 4
    for f in names:
        assert f"Simulated x rays:    {n}" in f.read_text(), f.name


def test_bench_line_keeps_the_driver_contract(tmp_path):
    """`python bench.py` prints ONE JSON line on stdout with the fields the driver and the judge read (metric / value / unit / n_gpus /
    steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline`
    {bound, achieved, peak, unit, frac, traffic} and `cpu_baseline` {value, unit, cores, kind, sample}; a reduced run here (small
    CPU budget, no scan legs) -- the line must parse, the check must pass, the exit code must be 0."""
    import json
    import sys
    r = subprocess.run([sys.executable, str(cases.ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-end-to-end", "--no-workloads", "--no-compat",
                        "--no-fdk", "--cpu-seconds", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "histories/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    # the asterisk on "f32" is resolved in the line itself: the same kernel with the reference's double-precision sub-steps
    assert "rotate_double" in d["dtype_note"] and 0.5 * d["value"] < d["value_reference_arithmetic"] < 1.02 * d["value"]
    assert d["reference_arithmetic"]["mode"] == "fast64"
    assert d["value"] > 1e9 and abs(d["value"] - 1e8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["traffic"] is not None and roof["binding"]["resource"] == "valu lane-slots", "the committed PMC summary must belong to this kernel build"
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 1e4
    assert d["measured_ceilings"]["scattered_64bit_atomic_adds_per_s"] > 5e9 and d["check"]["passed"] is True
