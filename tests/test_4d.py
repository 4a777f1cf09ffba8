"""4-D support: one resident context serving several (geometry, projection angle) jobs, against the
per-state file-based flow of the reference (cbctmc/mc/simulation.py:527-710)."""
import numpy as np
import pytest

import cases
import golden_util as gu
import warp_ref


def _slab():
    g = cases.geometry.MCBoxGeometry(shape=(24, 20, 16), image_spacing=(10.0, 10.0, 10.0), material="h2o")
    g.materials[6:14, 5:15, 4:12] = cases.materials.material_number("bone_050")
    g.densities[6:14, 5:15, 4:12] = 1.4
    return g


class _ShiftModel:
    """Stand-in correspondence model: rigid SI shift proportional to the signal (input 4)."""

    def __init__(self, shape, amplitude_voxels=3.4):
        self.shape, self.amp = shape, amplitude_voxels

    def predict(self, x):
        u = np.zeros((3,) + tuple(self.shape), dtype=np.float32)
        u[2] = self.amp * float(x[0])
        u[0] = 0.5 * float(x[1])
        return u


def test_set_projection_angles_equals_an_input_file_with_those_angles(engine, tmp_path):
    g = _slab()
    mats, spc = cases.material_files(), cases.spectrum_file()
    angles = [270.0, 270.0, 295.5, 10.25, 123.0]
    a = cases.simulation.MCSimulation(g, mats, spc, projection_angles=angles, n_histories=1000, **cases.SMALL_DET).prepare_simulation(tmp_path / "a")
    b = cases.simulation.MCSimulation(g, mats, spc, projection_angles=[0.0, 90.0], n_histories=1000, **cases.SMALL_DET).prepare_simulation(tmp_path / "b")
    with engine.create(a, device=-1) as ca, engine.create(b, device=-1) as cb:
        cb.set_projection_angles(angles)
        assert cb.num_projections == 5
        assert np.array_equal(ca.host_table("source_data"), cb.host_table("source_data"))
        assert np.array_equal(ca.host_table("detector_data"), cb.host_table("detector_data"))
        assert [ca.projection_file_name(p).split("/")[-1] for p in range(5)] == [cb.projection_file_name(p).split("/")[-1] for p in range(5)]
        with pytest.raises(engine.EngineError):
            cb.set_projection_angles([1.0])


def test_set_geometry_arrays_equals_a_fresh_context_on_the_file(engine, tmp_path):
    """Arrays handed over in-process give the host tables of the voxel-file flow, Woodcock majorant included."""
    rng = np.random.default_rng(8)
    g0, g1 = _slab(), _slab()
    g1.materials[2:5, 2:5, 2:5] = cases.materials.material_number("teflon")
    g1.densities[:] = rng.uniform(0.5, 2.5, g1.densities.shape).astype(np.float32)  # exercises the "%.6f" quantisation
    mats, spc = cases.material_files(), cases.spectrum_file()
    kw = dict(n_projections=2, angle_between_projections=90.0, n_histories=1000, **cases.SMALL_DET)
    f0 = cases.simulation.MCSimulation(g0, mats, spc, **kw).prepare_simulation(tmp_path / "g0")
    f1 = cases.simulation.MCSimulation(g1, mats, spc, **kw).prepare_simulation(tmp_path / "g1")
    with engine.create(f0, device=-1) as c0, engine.create(f1, device=-1) as c1:
        assert not np.array_equal(c0.host_table("mfp_woodcock"), c1.host_table("mfp_woodcock"))
        c0.set_geometry(g1)
        for name in ("voxel_mat_dens", "density_max", "mfp_woodcock", "mfp_a", "mfp_b", "fco", "noscco", "pmax"):
            assert np.array_equal(c0.host_table(name), c1.host_table(name)), name


def test_respiratory_signal_rules():
    """The rules that decide which projections share a warped geometry (cbctmc/mc/respiratory.py:45-93), stated independently here:
    resampling = linear interpolation onto int(T f) instants; quantisation = numpy.digitize's right-open classes mapped to their centres
    (so the maximum sits half a class ABOVE the range); grouping = np.unique(axis=0) order with ascending projection indices."""
    R = cases.pkg.respiratory.RespiratorySignal
    base = R.create_sin4(total_seconds=20.0, period=5.0, sampling_frequency=25.0)
    t25 = np.linspace(0, 20.0, 500)
    assert np.array_equal(base.signal, 1.0 * np.sin(2 * np.pi * (1 / (2 * 5.0)) * t25) ** 4) and np.array_equal(base.time, t25)
    assert np.array_equal(R.create_cos4(20.0, 5.0, 2.0, 25.0).signal, 2.0 * np.cos(2 * np.pi * (1 / (2 * 5.0)) * t25) ** 4)
    assert np.array_equal(base.dt_signal, np.gradient(base.signal, 1 / 25.0))
    s = base.resample(15.0)
    t15 = np.linspace(0, 20.0, 300)
    assert len(s.signal) == 300 and abs(s.signal.max() - 1.0) < 0.02
    assert np.array_equal(s.signal, np.interp(t15, t25, base.signal)) and np.array_equal(s.dt_signal, np.interp(t15, t25, base.dt_signal))
    for values, n in ((s.signal, 4), (s.dt_signal, 2), (np.array([0.0, 0.25, 0.5, 0.75, 1.0, 0.1]), 4)):
        q = R.quantize_signal(values, n_bins=n)
        edges = np.linspace(values.min(), values.max(), n + 1)
        want = edges[np.digitize(values, bins=edges) - 1] + 0.5 * (edges[1] - edges[0])
        assert np.array_equal(q, want) and len(np.unique(q)) <= n + 1
        assert q[np.argmax(values)] == values.max() + 0.5 * (edges[1] - edges[0])   # the maximum's own class
    q, dq = R.quantize_signal(s.signal, n_bins=4), R.quantize_signal(s.dt_signal, n_bins=2)
    u = R.get_unique_signals(q, dq)
    assert sorted(i for v in u.values() for i in v) == list(range(300))
    samples = np.stack((q, dq), axis=-1)
    want = {tuple(p.tolist()): np.where((samples == p).all(axis=1))[0].tolist() for p in np.unique(samples, axis=0)}
    assert list(u.items()) == list(want.items())   # same states, same order, same ascending indices


def test_warp_restatement_equals_torch_grid_sample_fixture():
    """tests/warp_ref.py (numpy) against the known answers torch itself produced (oracle/gen_warp_golden.py): ties, both
    borders of every axis within a few ulps, far-away samples, default values."""
    g = gu.load("warp_kat.npz")
    for k in range(int(g["n_cases"])):
        wm, wd = warp_ref.warp_nearest(g[f"materials_{k}"], g[f"densities_{k}"], g[f"field_{k}"], int(g["default_material"]), float(g["default_density"]))
        assert np.array_equal(wm, g[f"warped_materials_{k}"]) and np.array_equal(wd, g[f"warped_densities_{k}"]), k
    # the shortcut out[x] = in[rint(x + u)] is NOT the same function (ties and border samples move through the normalisation)
    m, d, u = g["materials_1"], g["densities_1"], g["field_1"]
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in m.shape], indexing="ij")).astype(np.float32)
    s = [np.rint(idx[c] + u[c]) for c in range(3)]
    inside = np.all([(s[c] >= 0) & (s[c] <= m.shape[c] - 1) for c in range(3)], axis=0)
    i = [np.clip(s[c], 0, m.shape[c] - 1).astype(int) for c in range(3)]
    assert np.count_nonzero(np.where(inside, m[i[0], i[1], i[2]], 1) != g["warped_materials_1"]) > 0


@pytest.mark.gpu
def test_gpu_warp_equals_torch_grid_sample_fixture(engine, case_dir):
    """csrc/warp.hip through the C ABI (and MCGeometry.warp on top of it) against the torch-generated known answers."""
    g = gu.load("warp_kat.npz")
    with engine.create(case_dir("air"), device=0) as ctx:
        for k in range(int(g["n_cases"])):
            m, d, u = g[f"materials_{k}"], g[f"densities_{k}"], g[f"field_{k}"]  # [x, y, z], field [3, x, y, z]
            mo, do = ctx.warp_volume(np.transpose(m, (2, 1, 0)), np.transpose(d, (2, 1, 0)), np.ascontiguousarray(np.transpose(u, (0, 3, 2, 1))),
                                     default_material=int(g["default_material"]), default_density=float(g["default_density"]))
            assert np.array_equal(np.transpose(mo, (2, 1, 0)), g[f"warped_materials_{k}"]), k
            assert np.array_equal(np.transpose(do, (2, 1, 0)), g[f"warped_densities_{k}"]), k
        geo = cases.geometry.MCGeometry(g["materials_3"], g["densities_3"], (2.0, 2.0, 2.0))
        # MCGeometry.warp: air (material 1, density 0.0013) outside, [1, 3, x, y, z] fields accepted like the reference
        w = geo.warp(g["field_3"][None], ctx)
        assert np.array_equal(w.materials, g["warped_materials_3"]) and np.array_equal(w.densities, g["warped_densities_3"])
        assert w.image_spacing == geo.image_spacing
        with pytest.raises(ValueError):
            geo.warp(g["field_2"], ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("case,records", [("cirs76", "0"), ("cirs76", "1"), ("slab4d", "0"), ("slab4d", "1"), ("thorax128_bone", "1")])
def test_device_geometry_warp_equals_the_host_route(engine, case_dir, tmp_path, monkeypatch, case, records):
    """mcgpu_warp_geometry (index volume warped, both brick levels -- with `records`, the 16-byte tile records rebuilt by the
    classify kernel --, object box and Woodcock majorant rebuilt on the device) against the route through the host: warp in the
    MCGeometry frame (tests/warp_ref.py, pinned by the torch fixture), a fresh context on the warped voxel file.  Same host
    tables, same COMPAT tallies bit for bit, same FAST tallies."""
    monkeypatch.setenv("MCGPU_TILE_RECORDS", records)
    mats, spc = cases.material_files(), cases.spectrum_file()
    if case in cases.CASES:  # thorax128_bone: thousands of tiles of three and four entries, encoded by the device builder after the warp
        g = cases.CASES[case][0]()
    else:
        g = _slab()
    rng = np.random.default_rng(5)
    shape = g.materials.shape
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n, dtype=np.float32) for n in shape], indexing="ij")
    field = np.stack([2.5 * np.sin(2.0 * y) + 0.5, 1.5 * x * z - 0.5, 3.0 * np.cos(1.5 * x) * (1 - z * z)]).astype(np.float32)
    field[:, ::7, ::5, ::3] += rng.uniform(-3, 3, size=field[:, ::7, ::5, ::3].shape).astype(np.float32)
    field[0, 1, :, :] = 0.5  # ties along the first axis
    kw = dict(n_projections=2, angle_between_projections=70.0, n_histories=200_000, **cases.SMALL_DET)
    base = cases.simulation.MCSimulation(g, mats, spc, **kw).prepare_simulation(tmp_path / "base")
    air = cases.materials.material_number("air")
    wm, wd = warp_ref.warp_nearest(g.materials, g.densities, field, air, cases.materials.MATERIALS_125KEV["air"])
    assert (wm != g.materials).sum() > 100
    warped = cases.simulation.MCSimulation(cases.geometry.MCGeometry(wm, wd, g.image_spacing), mats, spc, **kw).prepare_simulation(tmp_path / "warped")
    with engine.create(base, device=0) as dev, engine.create(warped, device=0) as ref:
        before = dev.host_table("mfp_woodcock").copy()
        dev.warp_geometry(field, frame="geometry")
        assert np.array_equal(dev.host_table("voxel_mat_dens"), ref.host_table("voxel_mat_dens"))  # downloaded on demand
        assert np.array_equal(dev.host_table("density_max").view("<f4")[:22] > 0, ref.host_table("density_max").view("<f4")[:22] > 0)
        assert np.array_equal(dev.host_table("mfp_woodcock"), ref.host_table("mfp_woodcock"))
        for key in ("bricks_mixed", "bricks_exterior", "brick_shift", "brick_count"):
            # "bricks_mixed" counts the bricks whose 4-bit code says "ask the volume": mixed ones AND homogeneous ones of a palette entry
            # without a code of its own (only entries that fill at least one whole brick get one).  A warped context keeps the BASE
            # volume's codes, a fresh one gives codes by the warped volume's bricks: where an entry fills a whole brick in only one of
            # the two (thin structures: the thorax's bone outline), the counts differ -- the tallies below may not
            if key != "bricks_mixed" or case != "thorax128_bone":
                assert dev.geti(key) == ref.geti(key), key
        for p in range(2):
            a, _, _ = dev.run_projection(p, 300, mode="compat", seed=5 + p, hpt=100)
            b, _, _ = ref.run_projection(p, 300, mode="compat", seed=5 + p, hpt=100)
            assert np.array_equal(a, b) and a.sum() > 0
            a, _, _ = dev.run_projection(p, 400_000, mode="fast", seed=9)
            b, _, _ = ref.run_projection(p, 400_000, mode="fast", seed=9)
            assert np.array_equal(a, b) and a.sum() > 0
        # a clone of the warped context downloads (and un-tiles) the device-resident voxels: same geometry, same tallies
        with dev.clone(0) as twin:
            assert np.array_equal(twin.host_table("voxel_mat_dens"), ref.host_table("voxel_mat_dens"))
            a, _, _ = twin.run_projection(1, 300_000, mode="fast", seed=4)
            b, _, _ = dev.run_projection(1, 300_000, mode="fast", seed=4)
            assert np.array_equal(a, b) and a.sum() > 0
        # a second state warps the BASE geometry again (not the warped one), and the identity field restores it
        dev.warp_geometry(np.zeros_like(field), frame="geometry")
        with engine.create(base, device=0) as fresh:
            assert np.array_equal(dev.host_table("voxel_mat_dens"), fresh.host_table("voxel_mat_dens"))
            assert np.array_equal(dev.host_table("mfp_woodcock"), before)
            a, _, _ = dev.run_projection(1, 300_000, mode="fast", seed=3)
            b, _, _ = fresh.run_projection(1, 300_000, mode="fast", seed=3)
            assert np.array_equal(a, b)
        # the engine-frame form of the same call: the field rotated like the voxel file (rot90 k=3: x_e = y_g, y_e = -x_g)
        fe = np.stack([np.rot90(field[1], k=3, axes=(0, 1)), -np.rot90(field[0], k=3, axes=(0, 1)), np.rot90(field[2], k=3, axes=(0, 1))])
        dev.warp_geometry(np.ascontiguousarray(np.transpose(fe, (0, 3, 2, 1))), frame="engine")
        differ = np.count_nonzero(dev.host_table("voxel_mat_dens").view("<f4") != ref.host_table("voxel_mat_dens").view("<f4"))
        assert differ < 0.01 * wm.size  # same warp up to ties and border samples on the mirrored axis


@pytest.mark.gpu
def test_device_warp_that_creates_an_exterior_names_its_background(engine, tmp_path):
    """A base geometry whose object box spans the whole brick grid (bone blocks in all eight corners: no EXTERIOR bricks, the
    palette slot of brick code 14 is unset) and a respiratory state that pulls the object inward (the corners leave the
    volume, a centre block is magnified): the device rebuild now codes the rim as EXTERIOR and must point code 14 at the
    background (water) -- palette entry 0 is the corner's bone.  Same tallies as a fresh context on the warped voxel file."""
    mats, spc = cases.material_files(), cases.spectrum_file()
    g = cases.geometry.MCBoxGeometry(shape=(32, 32, 24), image_spacing=(10.0, 10.0, 10.0), material="h2o")
    bone = cases.materials.material_number("bone_050")
    for sx in (slice(0, 4), slice(28, 32)):
        for sy in (slice(0, 4), slice(28, 32)):
            for sz in (slice(0, 4), slice(20, 24)):
                g.materials[sx, sy, sz] = bone
                g.densities[sx, sy, sz] = 1.4
    g.materials[14:18, 14:18, 10:14] = bone
    g.densities[14:18, 14:18, 10:14] = 1.4
    shape = g.materials.shape
    grid = np.meshgrid(*[np.arange(n, dtype=np.float32) for n in shape], indexing="ij")
    centre = [(n - 1) / 2.0 for n in shape]
    field = np.stack([0.5 * (c - x) for x, c in zip(grid, centre)]).astype(np.float32)  # out[x] = base[(x + c) / 2]
    kw = dict(n_projections=2, angle_between_projections=70.0, n_histories=200_000, **cases.SMALL_DET)
    base = cases.simulation.MCSimulation(g, mats, spc, **kw).prepare_simulation(tmp_path / "base")
    air = cases.materials.material_number("air")
    wm, wd = warp_ref.warp_nearest(g.materials, g.densities, field, air, cases.materials.MATERIALS_125KEV["air"])
    assert (wm[:8, :8, :6] != bone).all() and (wm == bone).sum() > 64  # corners gone, centre block magnified
    warped = cases.simulation.MCSimulation(cases.geometry.MCGeometry(wm, wd, g.image_spacing), mats, spc, **kw).prepare_simulation(tmp_path / "warped")
    with engine.create(base, device=0) as dev, engine.create(warped, device=0) as ref:
        assert dev.geti("bricks_exterior") == 0 and ref.geti("bricks_exterior") > 0
        dev.warp_geometry(field, frame="geometry")
        assert np.array_equal(dev.host_table("voxel_mat_dens"), ref.host_table("voxel_mat_dens"))
        for key in ("bricks_mixed", "bricks_exterior", "brick_shift", "brick_count"):
            assert dev.geti(key) == ref.geti(key), key
        for p in range(2):
            a, _, _ = dev.run_projection(p, 400_000, mode="fast", seed=9)
            b, _, _ = ref.run_projection(p, 400_000, mode="fast", seed=9)
            assert np.array_equal(a, b) and a.sum() > 0
        # and back: the identity field restores the base, whose exterior is empty again
        dev.warp_geometry(np.zeros_like(field), frame="geometry")
        assert dev.geti("bricks_exterior") == 0
        with engine.create(base, device=0) as fresh:
            a, _, _ = dev.run_projection(1, 300_000, mode="fast", seed=3)
            b, _, _ = fresh.run_projection(1, 300_000, mode="fast", seed=3)
            assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("phantom", ["slab", "cirs76"])
def test_4d_scan_equals_per_state_file_based_runs(engine, tmp_path, phantom):
    """The resident 4-D driver (BASELINE config 5 at reduced size: CIRS phantom + correspondence model + respiratory signal)
    writes, slice for slice, what separate per-state simulations (the reference's flow) give."""
    g = _slab() if phantom == "slab" else cases.CASES["cirs76"][0]()
    mats, spc = cases.material_files(), cases.spectrum_file()
    R = cases.pkg.respiratory.RespiratorySignal
    model = _ShiftModel(g.materials.shape)
    sim4d = cases.simulation.MCSimulation4D(model, g, mats, spc, n_histories=200_000, n_projections=12, frame_rate=15.0,
                                            angle_between_projections=30.0, **cases.SMALL_DET)
    signal = R.create_sin4(total_seconds=2.0, period=1.0, sampling_frequency=25.0)
    rep = sim4d.run_simulation(signal, 3, tmp_path / "out", engine, mode="fast")
    assert rep["projections"] == 12 and 2 <= rep["unique_states"] <= 9
    total = engine.stack_read(tmp_path / "out" / "projections_total.mha")
    assert total.shape == (12, 96, 231)
    # per-state reference flow: warp (numpy), write the voxel file, fresh context, duplicated first angle, FAST mode
    sig = signal.resample(15.0)
    s, ds = R.quantize_signal(sig.signal[:12], 3), R.quantize_signal(sig.dt_signal[:12], 3)
    planes = np.zeros((12, 3, 96, 231), dtype=np.float32)
    for k, ((sv, dsv), idx) in enumerate(R.get_unique_signals(s, ds).items()):
        u = model.predict(np.array([sv, dsv]))
        wm, wd = warp_ref.warp_nearest(g.materials, g.densities, u, cases.materials.material_number("air"), cases.materials.MATERIALS_125KEV["air"])
        angles = [270.0 + i * 30.0 for i in idx]
        one = cases.simulation.MCSimulation(cases.geometry.MCGeometry(wm, wd, g.image_spacing), mats, spc, n_histories=200_000,
                                            projection_angles=angles[0:1] + angles, angle_between_projections=30.0, **cases.SMALL_DET)
        inp = one.prepare_simulation(tmp_path / f"state{k}")
        with engine.create(inp, device=0) as ctx:
            for j, i in enumerate(idx):
                img, _, done = ctx.run_projection(j + 1, 200_000, mode="fast", seed=ctx.geti("seed"))
                planes[i] = ctx.finalize_host(img, done)
    for k, m in enumerate(("total", "unscattered", "scattered")):
        want = planes[:, k]
        want = np.where(want == 0, want[want > 0].min(), want)
        assert np.array_equal(engine.stack_read(tmp_path / "out" / f"projections_{m}.mha"), want), m
