"""4-D support: one resident context serving several (geometry, projection angle) jobs, against the
per-state file-based flow of the reference (cbctmc/mc/simulation.py:527-710)."""
import numpy as np
import pytest

import cases


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
    R = cases.pkg.respiratory.RespiratorySignal
    s = R.create_sin4(total_seconds=20.0, period=5.0, sampling_frequency=25.0).resample(15.0)
    assert len(s.signal) == 300 and abs(s.signal.max() - 1.0) < 0.02
    q = R.quantize_signal(s.signal, n_bins=4)
    assert len(np.unique(q)) <= 5
    u = R.get_unique_signals(q, R.quantize_signal(s.dt_signal, n_bins=2))
    assert sorted(i for v in u.values() for i in v) == list(range(300))


@pytest.mark.gpu
def test_gpu_warp_matches_numpy_nearest_neighbour(engine, case_dir):
    rng = np.random.default_rng(2)
    nz, ny, nx = 11, 13, 17
    m = rng.integers(1, 22, (nz, ny, nx)).astype(np.uint8)
    d = rng.uniform(0.001, 2.7, (nz, ny, nx)).astype(np.float32)
    u = rng.uniform(-4, 4, (3, nz, ny, nx)).astype(np.float32)
    u[:, 0, 0, :6] = [[0.5, 1.5, 2.5, -0.5, -1.5, 0.0]] * 3  # ties: round half to even
    with engine.create(case_dir("air"), device=0) as ctx:
        mo, do = ctx.warp_volume(m, d, u, default_material=1, default_density=0.0012)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    sx, sy, sz = (np.rint(x.astype(np.float32) + u[0]), np.rint(y.astype(np.float32) + u[1]), np.rint(z.astype(np.float32) + u[2]))
    inside = (sx >= 0) & (sx <= nx - 1) & (sy >= 0) & (sy <= ny - 1) & (sz >= 0) & (sz <= nz - 1)
    ix, iy, iz = [np.clip(a, 0, n - 1).astype(int) for a, n in ((sx, nx), (sy, ny), (sz, nz))]
    assert np.array_equal(mo, np.where(inside, m[iz, iy, ix], 1))
    assert np.array_equal(do, np.where(inside, d[iz, iy, ix], np.float32(0.0012)))


@pytest.mark.gpu
def test_4d_scan_equals_per_state_file_based_runs(engine, tmp_path):
    """The resident 4-D driver writes, slice for slice, what separate per-state simulations (the reference's flow) give."""
    g = _slab()
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
        x, y, z = np.meshgrid(*[np.arange(n) for n in g.materials.shape], indexing="ij")
        src = [np.rint(a.astype(np.float32) + u[c]) for c, a in enumerate((x, y, z))]
        inside = np.all([(src[c] >= 0) & (src[c] <= g.materials.shape[c] - 1) for c in range(3)], axis=0)
        sx, sy, sz = [np.clip(src[c], 0, g.materials.shape[c] - 1).astype(int) for c in range(3)]
        wm = np.where(inside, g.materials[sx, sy, sz], cases.materials.material_number("air")).astype(np.uint8)
        wd = np.where(inside, g.densities[sx, sy, sz], np.float32(cases.materials.MATERIALS_125KEV["air"])).astype(np.float32)
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
