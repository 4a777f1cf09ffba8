"""GPU parity tests (run with -m gpu on an MI355X): the HIP kernels, through the C ABI, against the CPU oracle.

COMPAT personality: integer tallies must be BIT-IDENTICAL to oracle/mcgpu_oracle.c (portable math).
FAST personality  : per-pixel agreement within 3 sigma of the (compound-)Poisson noise, the tolerance
                    BASELINE.json's north_star states.
"""
import numpy as np
import pytest

import oracle_lib as ol
import parity

pytestmark = pytest.mark.gpu

CASE_BATCHES = [("air", 256), ("water", 512), ("catphan64", 512), ("catphan64_ct", 256), ("slab_angles", 256), ("graded_u16", 256),
                ("graded_raw", 256), ("cirs76", 256), ("thorax64", 256), ("tissue22", 256), ("thorax128_bone", 256)]


@pytest.fixture(scope="module")
def gpu_engine(engine):
    return engine


def test_portable_math_bit_exact(gpu_engine, case_dir):
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(1e-9, 1.0, 4000), rng.uniform(1.0, 8.0, 2000), rng.uniform(-40.0, 0.5, 2000),
                        np.float32(rng.uniform(0, 1, 2000)).astype(np.float64) * 6.283185307179586])
    with gpu_engine.create(case_dir("air"), device=0) as ctx:
        lg, ex, sn, cs = ctx.kat_math(np.abs(x) + 1e-300)
        lib = ol.oracle()
        import ctypes as C
        for i, v in enumerate(np.abs(x) + 1e-300):
            s, c = C.c_double(), C.c_double()
            lib.oracle_pm_sincos(v, C.byref(s), C.byref(c))
            assert lg[i] == lib.oracle_pm_log(v)
            assert ex[i] == lib.oracle_pm_exp(v) or (np.isinf(ex[i]) and np.isinf(lib.oracle_pm_exp(v)))
            assert sn[i] == s.value and cs[i] == c.value


def test_compat_expf_bit_exact(gpu_engine, case_dir):
    """expf as the COMPAT kernel evaluates it == the oracle's gl_expf (which equals the host libm on every float,
    oracle/check_libm.c), incl. the underflow corner and subnormal results."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-110.0, 0.5, 200000), rng.uniform(-1e-3, 1e-3, 20000), rng.uniform(0.5, 89.5, 20000),
                        [-103.97, -103.5, -103.28, -103.2, -150.0, 88.7, 88.8, 0.0, -0.0, 0.5, -np.inf]]).astype(np.float32)
    lib = ol.oracle()
    want = np.array([lib.oracle_gl_expf(float(v)) for v in x], dtype=np.float32)
    with gpu_engine.create(case_dir("air"), device=0) as ctx:
        got = ctx.kat_expf(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_compat_lean_sqrt_and_quotient_are_the_ieee_ones(gpu_engine, case_dir):
    """compat_math.inc: cm_sqrtf / cm_divf drop the range scaling of the compiler's IEEE sequences; on the operand ranges of the
    electron-shell arithmetic they must return the same bits as sqrtf and / (both evaluated on the device)."""
    rng = np.random.default_rng(23)
    with gpu_engine.create(case_dir("air"), device=0) as ctx:
        # square root: every float of 600 binades-slices between 1 and 1e13 (dense in the mantissa), plus random radicands
        m = np.arange(0, 1 << 23, 7, dtype=np.uint32)
        for e in (127, 128, 131, 140, 150, 160, 169, 170):  # exponents of 1 .. 2^43
            x = ((np.uint32(e) << np.uint32(23)) | m).view(np.float32)
            assert np.array_equal(ctx.kat_f32(0, x).view(np.uint32), ctx.kat_f32(1, x).view(np.uint32)), e
        x = np.exp(rng.uniform(np.log(1.0), np.log(1e13), 2_000_000)).astype(np.float32)
        assert np.array_equal(ctx.kat_f32(0, x).view(np.uint32), ctx.kat_f32(1, x).view(np.uint32))
        # quotient: numerators 0 or 1e-3 .. 1e13 of either sign, divisors 1e6 .. 1e12
        n = (np.exp(rng.uniform(np.log(1e-3), np.log(1e13), 4_000_000)) * rng.choice([-1.0, 1.0], 4_000_000)).astype(np.float32)
        n[:1000] = 0.0
        n[1000:2000] = -0.0
        d = np.exp(rng.uniform(np.log(1e6), np.log(1e12), 4_000_000)).astype(np.float32)
        assert np.array_equal(ctx.kat_f32(2, n, d).view(np.uint32), ctx.kat_f32(3, n, d).view(np.uint32))
        # the expression itself against numpy's IEEE float32 arithmetic: fj0 (aux - u mc2) / (sqrt(aux + aux + u u) mc2)
        fj0 = rng.uniform(1e-3, 200.0, 2_000_000).astype(np.float32)
        u = np.exp(rng.uniform(np.log(1.0), np.log(9e4), 2_000_000)).astype(np.float32)
        aux = (rng.uniform(0.0, 1.0, 2_000_000) * 4.5e10).astype(np.float32)
        mc2 = np.float32(510998.918)
        want = (fj0 * (aux - u * mc2)) / (np.sqrt(aux + aux + u * u) * mc2)
        assert want.dtype == np.float32
        assert np.array_equal(ctx.kat_f32(4, fj0, aux, u).view(np.uint32), want.view(np.uint32))


def test_ranecu_stream_bit_exact(gpu_engine, case_dir):
    import ctypes as C
    with gpu_engine.create(case_dir("air"), device=0) as ctx:
        for batch, hpt, seed in [(0, 150, 42), (1, 150, 42), (66666, 1431, 1267439713), (123456, 7, 99)]:
            got = ctx.kat_rng("compat", seed, batch, hpt, 2000)
            s = (C.c_int * 2)()
            ol.oracle().oracle_init_prng(batch, hpt, seed, s)
            want = np.array([ol.oracle().oracle_ranecu(s) for _ in range(2000)], dtype=np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name,nbatch", CASE_BATCHES)
def test_compat_kernel_bit_exact_vs_oracle(gpu_engine, case_dir, name, nbatch):
    with gpu_engine.create(case_dir(name), device=0) as ctx:
        T = parity.tables_from_context(ctx)
        for p in range(ctx.num_projections):
            seed = 42 + 1000 * p
            img_gpu, secs, done = ctx.run_projection(p, nbatch, mode="compat", seed=seed, hpt=150)
            img_cpu, _ = T.track(p, seed, 0, nbatch, 150, ol.MATH_PORTABLE, n_threads=4)
            assert done == nbatch * 150
            assert img_gpu.sum() > 0
            diff = np.count_nonzero(img_gpu.reshape(-1) != img_cpu)
            assert diff == 0, f"{name} projection {p}: {diff} tally words differ"


@pytest.mark.parametrize("name", [c for c, _ in CASE_BATCHES] + ["catphan64_dose"])
def test_compat_kernel_against_the_reference_build_itself(gpu_engine, case_dir, name):
    """Closes the chain GPU -> portable oracle -> libm oracle -> reference ON THE GPU BOX: tests/golden/case_*.npz holds the tallies
    of the reference's own C code compiled where it lies (`ref_*`, oracle/gen_golden.py from oracle/_ref) and of the portable
    restatement (`portable_*`) for the same batches.  The COMPAT kernel must equal the portable fixture word for word and may
    differ from the REFERENCE BUILD only where the portable logarithm / power / sine differ from glibc's in the last bit
    (pm_log vs logf: 4.2e5 of 2.1e9 floats) -- the budget tests/test_oracle_golden.py states for the two CPU modes."""
    import golden_util as gu
    g = gu.load(f"case_{name}.npz")
    nb, hpt = [int(v) for v in g["nbatch_hpt"]]
    with gpu_engine.create(case_dir(name), device=0) as ctx:
        size = int(np.prod(ctx.detector_shape)) * 4
        for p in range(ctx.num_projections):
            img, _, done = ctx.run_projection(p, nb, mode="compat", seed=42 + 1000 * p, hpt=hpt)
            assert done == nb * hpt
            got = img.reshape(-1)
            assert np.array_equal(got, gu.dense(g, "portable", p, size)), f"{name} p{p}: differs from the portable-math fixture"
            ref = gu.dense(g, "ref", p, size)
            ndiff = np.count_nonzero(got != ref)
            assert ndiff <= max(8, np.count_nonzero(ref) // 500), f"{name} p{p}: {ndiff} tally words differ from the reference build"
            # the words that differ come in pairs (a history scored one pixel over, or with one unit less): energy is conserved
            assert abs(int(got.sum()) - int(ref.sum())) <= 1e-6 * int(ref.sum())


def test_compat_history_sharding_is_exact(gpu_engine, case_dir):
    """Two ranks given disjoint batch ranges sum to the single-GPU image (what the multi-GPU reduce relies on)."""
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        whole, _, _ = ctx.run_projection(0, 300, mode="compat", seed=42, hpt=100)
        a, _, _ = ctx.run_projection(0, 130, mode="compat", seed=42, hpt=100, first=0)
        b, _, _ = ctx.run_projection(0, 170, mode="compat", seed=42, hpt=100, first=130)
        assert np.array_equal(whole, a + b)


@pytest.mark.parametrize("name", ["tissue22", "catphan64"])
def test_compat_image_is_independent_of_its_batching(gpu_engine, case_dir, monkeypatch, name):
    """The COMPAT kernel runs two RANECU batches per lane and serves Compton / Rayleigh / tally batches at ballot thresholds:
    none of that may reach the tallies -- every threshold setting, an odd number of batches (one lane with a single batch), a
    single batch, and a launch split in two give the same words."""
    with gpu_engine.create(case_dir(name), device=0) as ctx:
        ref, _, _ = ctx.run_projection(0, 301, mode="compat", seed=42, hpt=60)
        for cfg in ("1,1,1,1", "64,64,64,64", "48,4,12,4", "20,8,40,32", "33,2,7,64"):
            for k, v in zip(("MCGPU_COMPAT_THRESH_COMPTON", "MCGPU_COMPAT_THRESH_RAYLEIGH", "MCGPU_COMPAT_THRESH_NEW", "MCGPU_COMPAT_THRESH_TAKE"), cfg.split(",")):
                monkeypatch.setenv(k, v)
            ctx.reload_env_knobs()
            img, _, _ = ctx.run_projection(0, 301, mode="compat", seed=42, hpt=60)
            assert np.array_equal(img, ref), cfg
        a, _, _ = ctx.run_projection(0, 1, mode="compat", seed=42, hpt=60, first=0)
        b, _, _ = ctx.run_projection(0, 300, mode="compat", seed=42, hpt=60, first=1)
        assert a.sum() > 0 and np.array_equal(a + b, ref)
        T = parity.tables_from_context(ctx)
        cpu, _ = T.track(0, 42, 0, 301, 60, ol.MATH_PORTABLE, n_threads=4)
        assert np.array_equal(ref.reshape(-1), cpu)
    for k in ("MCGPU_COMPAT_THRESH_COMPTON", "MCGPU_COMPAT_THRESH_RAYLEIGH", "MCGPU_COMPAT_THRESH_NEW", "MCGPU_COMPAT_THRESH_TAKE"):
        monkeypatch.delenv(k, raising=False)


def test_fast_history_sharding_and_determinism(gpu_engine, case_dir):
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        n = 400_000
        whole, _, done = ctx.run_projection(0, n, mode="fast", seed=7)
        again, _, _ = ctx.run_projection(0, n, mode="fast", seed=7)
        assert done == n
        assert np.array_equal(whole, again), "integer tallies must not depend on scheduling"
        a, _, _ = ctx.run_projection(0, 150_000, mode="fast", seed=7, first=0)
        b, _, _ = ctx.run_projection(0, 250_000, mode="fast", seed=7, first=150_000)
        assert np.array_equal(whole, a + b)
        other, _, _ = ctx.run_projection(0, n, mode="fast", seed=8)
        assert not np.array_equal(whole, other)


def test_volume_storage_kinds_are_exercised(gpu_engine, case_dir):
    """u8 palette / u16 palette / raw float2 voxels (device_model.hpp): the graded cases reach the two wider kinds."""
    kinds = {}
    for name in ("catphan64", "graded_u16", "graded_raw"):
        with gpu_engine.create(case_dir(name), device=0) as ctx:
            kinds[name] = ctx.geti("volume_kind")
    assert kinds == {"catphan64": 0, "graded_u16": 1, "graded_raw": 2}


def test_lds_image_keeps_two_workgroups_per_cu_with_all_22_materials(gpu_engine, case_dir):
    """458 Compton shells (22 materials; blood alone has MAX_SHELLS = 40) take 7.3 KB of LDS against 1.4 KB for the Catphan
    set: the host sizes the brick grid and the bracket table so that the FAST kernel's image stays within 80 KB."""
    for name, nmat in (("tissue22", 22), ("thorax64", 14), ("cirs76", 7)):
        with gpu_engine.create(case_dir(name), device=0) as ctx:
            assert ctx.geti("num_materials_used") == nmat
            ctx.run_projection(0, 200_000, mode="fast", seed=1)
            assert ctx.geti("lds_bytes_fast") <= 80 * 1024 and ctx.geti("blocks_per_cu") == 2, name
            assert ctx.geti("sigma_bracket_shift") >= 6, name


FAST_CASES = ["catphan64", "water", "air", "slab_angles", "graded_u16", "graded_raw", "cirs76", "thorax64", "tissue22", "thorax128_bone"]


@pytest.mark.parametrize("name,mode", [(n, "fast") for n in FAST_CASES] + [(n, "fast64") for n in ("catphan64", "slab_angles", "cirs76", "thorax64", "tissue22")])
def test_fast_kernel_within_3_sigma_of_oracle(gpu_engine, case_dir, name, mode):
    """FAST vs oracle (LIBM math = the reference's own arithmetic): every class image, per pixel.  mode "fast64": the same kernel
    with the reference's three double-precision sub-steps (rotate_double, GRAa, GCOa's cdt1 / costh chain: csrc/track_fast64.hip)."""
    with gpu_engine.create(case_dir(name), device=0) as ctx:
        T = parity.tables_from_context(ctx)
        p = ctx.num_projections - 1
        nb, hpt = 4000, 150  # 6e5 oracle histories
        img_cpu, w2_cpu, _ = T.track_with_variance(p, 42, 0, nb, hpt, ol.MATH_LIBM, n_threads=8)
        n_cpu = nb * hpt
        n_gpu = 20_000_000
        img_gpu, secs, done = ctx.run_projection(p, n_gpu, mode=mode, seed=42)
        img_cpu, w2_cpu = img_cpu.reshape(img_gpu.shape), w2_cpu.reshape(img_gpu.shape)
        # integral quantities: detected energy per history, per image class, 3.5 sigma of the measured variance
        zs = parity.class_energy_z(img_gpu, done, img_cpu, w2_cpu, n_cpu)
        assert sum(np.isfinite(zs)) >= 1
        for k, zk in enumerate(zs):
            assert not np.isfinite(zk) or abs(zk) < 3.5, f"{name} class {k}: z = {zk:.2f} ({img_gpu[k].sum() / done:.6g} vs {img_cpu[k].sum() / n_cpu:.6g} per history)"
        # per pixel, coarse-grained 3x3 so that oracle pixels hold enough hits
        z, mask = parity.measured_z(parity.blocks(img_gpu), done, parity.blocks(img_cpu), parity.blocks(w2_cpu), n_cpu)
        assert mask.sum() > 50
        frac3 = np.mean(np.abs(z[mask]) > 3.0)
        assert frac3 < 0.01, f"{name}: {frac3:.4f} of {mask.sum()} blocks beyond 3 sigma (expect ~0.003)"
        assert np.abs(z[mask]).max() < 6.0
        assert abs(z[mask].mean()) < 0.25


def test_fast_image_is_independent_of_the_schedule(gpu_engine, case_dir, monkeypatch):
    """Per-history RNG streams + integer tallies: batching thresholds, exchange cadence and the exterior hop's timing must not
    change a single tally word (this is also what makes the multi-GPU sum exact).  Catches histories dropped or duplicated by
    the wave scheduler."""
    with gpu_engine.create(case_dir("catphan64_ct"), device=0) as ctx:
        n = 1_500_000
        ref = [ctx.run_projection(p, n, mode="fast", seed=11)[0] for p in (0, 2)]
        for knobs in ({"MCGPU_THRESH_COMPTON": "1", "MCGPU_THRESH_RAYLEIGH": "1", "MCGPU_THRESH_NEW": "1", "MCGPU_SWAP_BATCH": "1"},
                      {"MCGPU_THRESH_COMPTON": "64", "MCGPU_THRESH_RAYLEIGH": "64", "MCGPU_THRESH_NEW": "64", "MCGPU_FLYABLE_LOW": "1", "MCGPU_SWAP_BATCH": "40"},
                      {"MCGPU_THRESH_COMPTON": "7", "MCGPU_THRESH_NEW": "50", "MCGPU_FLYABLE_LOW": "40", "MCGPU_SWAP_BATCH": "3"}):
            for k, v in knobs.items():
                monkeypatch.setenv(k, v)
            ctx.reload_env_knobs()
            for i, p in enumerate((0, 2)):
                for rep in range(2):
                    img = ctx.run_projection(p, n, mode="fast", seed=11)[0]
                    assert np.array_equal(img, ref[i]), (knobs, p, rep)
            for k in knobs:
                monkeypatch.delenv(k)


@pytest.mark.parametrize("case", ["thorax64", "tissue22", "thorax128_bone"])
def test_fast_image_is_independent_of_slot_trading_segment_rule_and_brick_levels(gpu_engine, case_dir, monkeypatch, case):
    """Round-2 scheduler and lookup features, same statement as above -- identical tally WORDS required:
    slot trading (MCGPU_SLOT_TRADE 0..3: lanes re-point their LDS slots with six cross-lane permutes, the place where a
    history could be dropped or duplicated), the segment-end rule (MCGPU_HOLD_Q), the second brick level (MCGPU_SUB_BRICKS)
    and the size of the first one (MCGPU_MAX_BRICKS: coarser bricks turn homogeneous lookups into mixed ones and move the
    object box).  Tissue volumes, where mixed bricks dominate."""
    n, p = 1_200_000, 1
    with gpu_engine.create(case_dir(case), device=0) as ctx:
        ref, _, done = ctx.run_projection(p, n, mode="fast", seed=21)
        assert done == n and ref.sum() > 0
        for trade in (0, 1, 2, 3):
            for hold_q in (0, 6, 15):
                monkeypatch.setenv("MCGPU_SLOT_TRADE", str(trade))
                monkeypatch.setenv("MCGPU_HOLD_Q", str(hold_q))
                ctx.reload_env_knobs()
                img = ctx.run_projection(p, n, mode="fast", seed=21)[0]
                assert np.array_equal(img, ref), (trade, hold_q)
        monkeypatch.delenv("MCGPU_SLOT_TRADE")
        monkeypatch.delenv("MCGPU_HOLD_Q")
        # the flight segment as an inner loop or as part of the main loop (kSegmentLoop, chosen by the host from the materials): both
        # template variants on the same case, whichever the heuristic picks
        for seg in ("0", "1"):
            monkeypatch.setenv("MCGPU_SEGMENT_LOOP", seg)
            ctx.reload_env_knobs()
            assert ctx.geti("segment_loop") == int(seg)
            assert np.array_equal(ctx.run_projection(p, n, mode="fast", seed=21)[0], ref), ("segment loop", seg)
        monkeypatch.delenv("MCGPU_SEGMENT_LOOP")
        ctx.reload_env_knobs()
    # The brick SIZE moves the object box, and with it the region the exterior hop crosses analytically: a different (equally
    # valid) use of the random numbers.  With the hop off, a lookup returns the same (material, density) whatever the two
    # brick levels look like, so the tallies may not move at all; with it on, the second level alone may not move them.
    seen, compat_ref, plain_ref = set(), None, None
    for no_exterior in (False, True):
        if no_exterior:
            monkeypatch.setenv("MCGPU_NO_EXTERIOR", "1")
        for sub in ("0", "1", "records"):  # second level off / 4-bit codes / 16-byte tile records (MCGPU_TILE_RECORDS)
            for max_bricks in ((None, "300", "40") if no_exterior else (None,)):
                monkeypatch.setenv("MCGPU_SUB_BRICKS", "1" if sub == "1" else "0")
                monkeypatch.setenv("MCGPU_TILE_RECORDS", "1" if sub == "records" else "0")
                if max_bricks:
                    monkeypatch.setenv("MCGPU_MAX_BRICKS", max_bricks)
                else:
                    monkeypatch.delenv("MCGPU_MAX_BRICKS", raising=False)
                with gpu_engine.create(case_dir(case), device=0) as ctx:
                    assert (ctx.geti("bricks_exterior") == 0) == no_exterior
                    if max_bricks:
                        assert ctx.geti("brick_count") <= int(max_bricks)
                    assert ctx.geti("tile_records") == (1 if sub == "records" else 0)
                    img = ctx.run_projection(p, n, mode="fast", seed=21)[0]
                    if no_exterior:
                        seen.add((ctx.geti("brick_shift"), ctx.geti("bricks_mixed"), ctx.geti("sub_bricks_mixed")))
                        plain_ref = img if plain_ref is None else plain_ref
                        assert np.array_equal(img, plain_ref) and img.sum() > 0, (sub, max_bricks)
                    else:
                        assert np.array_equal(img, ref), (sub, max_bricks)
                    # COMPAT reads the same brick grid (no hop): its tallies may not move either
                    a = ctx.run_projection(p, 256, mode="compat", seed=5, hpt=50)[0]
                    compat_ref = a if compat_ref is None else compat_ref
                    assert np.array_equal(a, compat_ref) and a.sum() > 0, (no_exterior, sub, max_bricks)
    assert len({s[0] for s in seen}) == 3  # three brick sizes were really exercised


def test_fast_cross_section_brackets_decide_like_the_exact_table(gpu_engine, case_dir, monkeypatch):
    """The flight step decides most virtual/real tests from the LDS brackets of the total cross section (track_pool.inc:
    flight_step) and fetches the exact value only inside the bracket: the decisions -- hence every tally word -- must be
    those of the exact table alone (MCGPU_NO_BRACKETS)."""
    imgs = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("MCGPU_NO_BRACKETS", "1")
        for case in ("catphan64_ct", "graded_u16"):
            with gpu_engine.create(case_dir(case), device=0) as ctx:
                assert (ctx.geti("sigma_bracket_shift") < 0) == off
                imgs[(case, off)] = ctx.run_projection(1, 1_000_000, mode="fast", seed=3)[0]
    for case in ("catphan64_ct", "graded_u16"):
        assert np.array_equal(imgs[(case, False)], imgs[(case, True)]), case
        assert imgs[(case, False)].sum() > 0


def test_fast_exterior_hop_is_statistically_equivalent_to_delta_tracking(gpu_engine, case_dir, monkeypatch):
    """The analytic crossing of the homogeneous exterior (track_pool.inc: exterior_hop) against plain Woodcock tracking
    everywhere (MCGPU_NO_EXTERIOR): two independent estimates of the same images."""
    n = 30_000_000
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        assert ctx.geti("bricks_exterior") > 0
        hop, _, _ = ctx.run_projection(0, n, mode="fast", seed=5)
    monkeypatch.setenv("MCGPU_NO_EXTERIOR", "1")
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        assert ctx.geti("bricks_exterior") == 0
        plain, _, _ = ctx.run_projection(0, n, mode="fast", seed=6)
    # mean-square tally weight per image class, measured on an oracle sample (the GPU kernels tally no squares)
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        T = parity.tables_from_context(ctx)
    img_o, w2_o, _ = T.track_with_variance(0, 42, 0, 4000, 150, ol.MATH_LIBM, n_threads=8)
    img_o, w2_o = img_o.reshape(hop.shape), w2_o.reshape(hop.shape)
    r = np.array([w2_o[k].sum() / max(img_o[k].sum(dtype=np.float64), 1.0) for k in range(4)])[:, None, None]
    for k in range(4):
        z, m = parity.measured_z(np.array([hop[k].sum(dtype=np.float64)]), n, np.array([plain[k].sum(dtype=np.float64)]),
                                 np.array([plain[k].sum(dtype=np.float64) * r[k, 0, 0]]), n, min_hits=200.0)
        assert m[0] and abs(z[0]) < 4.0, (k, z)
    g, c = parity.blocks(hop), parity.blocks(plain)
    z, mask = parity.measured_z(g, n, c, c.astype(np.float64) * r, n, w2_a=g.astype(np.float64) * r)
    assert mask.sum() > 500
    assert np.mean(np.abs(z[mask]) > 3.0) < 0.01 and np.abs(z[mask]).max() < 6.0 and abs(z[mask].mean()) < 0.1


def test_fast_object_region_with_and_without_the_elliptic_cylinder(gpu_engine, case_dir, monkeypatch):
    """The object region of the exterior hop is the bounding box of everything that is not background AND, where it removes at least
    5 % of the box's bricks, an elliptic cylinder around it (model_device.cpp: mark_exterior_region; MCGPU_NO_ELLIPSE: box alone).
    Both regions hold the whole object, so the two runs are independent estimates of the same images; the cylinder must be in use
    on a round phantom and must mark more bricks exterior than the box."""
    n = 20_000_000
    res = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("MCGPU_NO_ELLIPSE", "1")
        for case in ("catphan64", "thorax64"):
            with gpu_engine.create(case_dir(case), device=0) as ctx:
                res[(case, off)] = (ctx.geti("exterior_cylinder"), ctx.geti("bricks_exterior"), ctx.run_projection(0, n, mode="fast", seed=11 + off)[0])
    assert any(res[(case, False)][0] == 1 for case in ("catphan64", "thorax64"))
    for case in ("catphan64", "thorax64"):
        (cyl, ext, img), (cyl0, ext0, img0) = res[(case, False)], res[(case, True)]
        assert cyl0 == 0 and (ext > ext0 if cyl else ext == ext0), (case, cyl, ext, ext0)
        # A sum S of tally weights w (unit 0.01 eV, w <= 1.5e7) has variance sum(w^2) <= 1.5e7 S: a bound on the sigma of every
        # class energy and block, somewhat above the true one (the mean photon carries ~6e6 units)
        for k in range(4):
            a, b = img[k].sum(dtype=np.float64), img0[k].sum(dtype=np.float64)
            if min(a, b) < 1e10:
                continue
            assert abs(a / b - 1.0) < 5.0 * np.sqrt(2.0 * 1.5e7 / min(a, b)), (case, k, a / b - 1.0)
        g, c = parity.blocks(img)[0].astype(np.float64), parity.blocks(img0)[0].astype(np.float64)
        m = (g > 1e10) & (c > 1e10)  # blocks of the primary image with well over a thousand photons
        z = (g[m] - c[m]) / np.sqrt((g[m] + c[m]) * 1.5e7)
        assert m.sum() > 20 and abs(z.mean()) < 5.0 / np.sqrt(m.sum()) and np.abs(z).max() < 6.0, (case, int(m.sum()), z.mean(), np.abs(z).max())


@pytest.mark.parametrize("mode", ["fast", "fast64"])
@pytest.mark.parametrize("case,projection,fixture", [("catphan64", 0, "stat_catphan64.npz"), ("slab_angles", 1, "stat_slab_angles_p1.npz")])
def test_fast_kernel_against_the_statistical_reference(gpu_engine, case_dir, case, projection, fixture, mode):
    """FAST vs tests/golden/stat_*.npz: 16 independent runs of the CPU oracle in its
    reference-identical mode -> per-block mean and run-to-run variance (catphan64: 16 x 4e7 histories, straight projection;
    slab_angles projection 1: 16 x 2.5e7, rotated source/detector at 300.5 degrees through water, bone and Teflon, i.e. the
    many-shell Compton class and the rotation branches of source and tally).  With as many GPU histories the comparison
    resolves a bias of a few hundredths of a sigma per 3x3-pixel block; tolerance: Student-t tails of a 16-run variance
    estimate."""
    import golden_util as gu
    g = gu.load(fixture)
    mean, var_ref = g["mean"].astype(np.float64), g["var_of_mean"].astype(np.float64)
    n_ref = int(g["histories_per_run"]) * int(g["runs"])
    n_gpu = n_ref
    with gpu_engine.create(case_dir(case), device=0) as ctx:
        img = np.zeros((4,) + ctx.detector_shape, dtype=np.float64)
        for k in range(4):  # four launches with different seeds: independent histories
            part, _, done = ctx.run_projection(projection, n_gpu // 4, mode=mode, seed=100 + k)
            img += part
        nz, nx = img.shape[1:]
        b = img[:, : nz // 3 * 3, : nx // 3 * 3].reshape(4, nz // 3, 3, nx // 3, 3).sum(axis=(2, 4)) / (n_gpu // 4 * 4)
    mask = var_ref > 0
    mask &= mean * n_ref / 6.0e6 > 50  # at least ~50 detected photons behind the reference mean
    z = (b[mask] - mean[mask]) / np.sqrt(var_ref[mask] * (1.0 + n_ref / n_gpu))
    assert mask.sum() > 2000
    assert np.mean(np.abs(z) > 3.0) < 0.02 and np.abs(z).max() < 8.0, (np.mean(np.abs(z) > 3.0), np.abs(z).max())
    assert abs(z.mean()) < 5.0 / np.sqrt(mask.sum()), z.mean()  # no systematic offset
    for k in range(4):  # detected energy per history per class
        zk = (b[k].sum() - mean[k].sum()) / np.sqrt(var_ref[k].sum() * 2.0)
        assert abs(zk) < 4.5, (k, zk)


@pytest.mark.parametrize("mode", ["fast", "fast64"])
def test_fast_kernel_tallies_are_pinned(gpu_engine, mode):
    """The production kernel is built without implicit FMA contraction, so its integer tallies are a function of the source
    alone: tests/golden/fast_pin.json (written on an MI355X by tools/gen_fast_pin.py) holds their SHA-256 on four small cases
    (fast64_pin.json: the variant with the reference's double-precision sub-steps).
    A mismatch means the FAST arithmetic or its random-number use changed -- regenerate the pin only if that was intended
    (the statistical tests above are what shows a change is still correct)."""
    import json
    import sys
    sys.path.insert(0, str(cases_root() / "tools"))
    import gen_fast_pin
    want = json.loads((cases_root() / "tests" / "golden" / f"{mode}_pin.json").read_text())
    got = gen_fast_pin.compute(mode)
    for name in want:
        assert got[name]["sum"] == want[name]["sum"] and got[name]["sha256"] == want[name]["sha256"], name


def test_fast64_differs_from_fast_only_where_the_reference_computes_in_double(gpu_engine, case_dir):
    """The two arithmetics share scheduler, streams and every float32 step: a geometry in which nothing scatters (air: one primary
    per history, no Compton / Rayleigh event reaches a rotation) gives the SAME tally words, and a scattering one differs -- but only
    in its scatter classes' details, not in the number of histories or, beyond a few 1e-4, in its class energies."""
    with gpu_engine.create(case_dir("catphan64"), device=0) as ctx:
        a, _, da = ctx.run_projection(0, 2_000_000, mode="fast", seed=5)
        b, _, db = ctx.run_projection(0, 2_000_000, mode="fast64", seed=5)
        assert da == db == 2_000_000 and not np.array_equal(a, b)
        ea, eb = a.reshape(4, -1).sum(axis=1).astype(np.float64), b.reshape(4, -1).sum(axis=1).astype(np.float64)
        # the same histories up to their first scattering: the primary image differs only through photons whose FIRST interaction's
        # outcome (a rotation) is not part of it at all -- it is identical
        assert np.array_equal(a[0], b[0])
        assert np.all(np.abs(eb[1:] / ea[1:] - 1.0) < 0.02)
        # determinism of the double-precision variant itself, and its independence of the schedule
        c, _, _ = ctx.run_projection(0, 2_000_000, mode="fast64", seed=5)
        assert np.array_equal(b, c)


def test_workgroup_level_scheduler_gives_the_pinned_tallies(gpu_engine, case_dir, monkeypatch):
    """MCGPU_FAST_SCHED=1 runs the same physics under the workgroup-level pool (track_pool.inc: track_wg_kernel -- histories parked in
    slots that belong to the workgroup, rings of slot ids per kind of work, full batches): measured slower than the per-wave pools
    (profiles/r05*_wg_*), kept as an opt-in.  One RNG stream per history and integer tallies: its images are the default
    scheduler's bit for bit -- the pinned digests -- including launches too small to fill a workgroup and ids beyond 2^32."""
    import json
    import sys
    sys.path.insert(0, str(cases_root() / "tools"))
    import gen_fast_pin
    want = json.loads((cases_root() / "tests" / "golden" / "fast_pin.json").read_text())
    with gpu_engine.create(case_dir("water"), device=0) as ctx:
        assert ctx.geti("fast_scheduler") == 0
        ref = {n: ctx.run_projection(0, n, mode="fast", seed=9, first=2 ** 32 - 50)[0] for n in (1, 63, 1000, 70_001)}
    monkeypatch.setenv("MCGPU_FAST_SCHED", "1")
    got = gen_fast_pin.compute()
    for name in want:
        assert got[name]["sum"] == want[name]["sum"] and got[name]["sha256"] == want[name]["sha256"], name
    with gpu_engine.create(case_dir("water"), device=0) as ctx:
        assert ctx.geti("fast_scheduler") == 1
        for n, img in ref.items():
            again, _, done = ctx.run_projection(0, n, mode="fast", seed=9, first=2 ** 32 - 50)
            assert done == n and np.array_equal(again, img), n


def cases_root():
    from pathlib import Path
    return Path(__file__).resolve().parents[1]


def test_fast_kernel_history_ids_beyond_32_bits(gpu_engine, case_dir):
    """BASELINE config 3 asks for 1.19e10 histories per projection: history ids (the counter the per-history seeding hashes), id ranges of a rank and
    the launch size itself pass 2^32.  A range that straddles 2^32 equals the sum of its two halves; a 5e9-history launch
    simulates exactly that many and agrees with 50 x a 1e8-history launch within the statistics of the latter."""
    with gpu_engine.create(case_dir("water"), device=0) as ctx:
        lo = 2 ** 32 - 70_000
        whole, _, done = ctx.run_projection(0, 150_000, mode="fast", seed=3, first=lo)
        a, _, da = ctx.run_projection(0, 70_000, mode="fast", seed=3, first=lo)
        b, _, db = ctx.run_projection(0, 80_000, mode="fast", seed=3, first=2 ** 32)
        assert done == 150_000 and da + db == done and np.array_equal(a + b, whole) and b.sum() > 0
        low, _, _ = ctx.run_projection(0, 80_000, mode="fast", seed=3, first=0)
        assert not np.array_equal(low, b)  # ids 2^32 + k are not ids k
        small, _, ds = ctx.run_projection(0, 100_000_000, mode="fast", seed=9)
        big, _, dbig = ctx.run_projection(0, 5_000_000_000, mode="fast", seed=9, first=2 ** 33)
        assert ds == 100_000_000 and dbig == 5_000_000_000
        for k in range(4):
            es, eb = float(small[k].sum()), float(big[k].sum())
            hits = max(es / 5.0e6, 1.0)  # ~ number of detected photons of the small run (tally unit 0.01 eV, ~50 keV each)
            assert abs(eb / 50.0 - es) < 6.0 * es / np.sqrt(hits) + 1.0, (k, es, eb)


def test_fast_kernel_small_and_ragged_history_counts(gpu_engine, case_dir):
    """History ids are dealt from 64 counters that each own count/64 ids (one more for the first count % 64): launches of 0, 1,
    63, 64, 65, ... histories simulate exactly that many, every id once -- the tallies of id ranges add up to the whole."""
    with gpu_engine.create(case_dir("water"), device=0) as ctx:
        empty, _, done = ctx.run_projection(0, 0, mode="fast", seed=3)
        assert done == 0 and int(empty.sum()) == 0
        whole, _, done = ctx.run_projection(0, 1000, mode="fast", seed=3)
        assert done == 1000 and whole.sum() > 0
        acc = np.zeros_like(whole)
        first = 0
        for n in (1, 63, 64, 65, 2, 128, 300, 377):
            part, _, d = ctx.run_projection(0, n, mode="fast", seed=3, first=first)
            assert d == n
            acc += part
            first += n
        assert first == 1000 and np.array_equal(acc, whole)
        # the same through the asynchronous ABI with the grid of a much larger launch in between (counters are reset per launch)
        big, _, _ = ctx.run_projection(0, 3_000_000, mode="fast", seed=3)
        again, _, _ = ctx.run_projection(0, 1000, mode="fast", seed=3)
        assert np.array_equal(again, whole) and big.sum() > whole.sum()


def test_fast64_double_precision_helpers_against_numpy(gpu_engine, case_dir):
    """csrc/track_fast64.hip's own arithmetic, held to double-precision numpy directly (the statistical tests cannot see an azimuth that is
    off by a quarter turn: scattering is symmetric about the photon's direction): sincos_turn over all four quadrants and their
    borders, rsqrt_d / sqrt_ratio_d / the refined quotient of compton_cdt1 to a few ulp of DOUBLE, and rotate_double
    (MC-GPU_kernel_v1.3.cu:1103-1148) restated in numpy float64 to one ulp of the float32 result."""
    rng = np.random.default_rng(3)
    n = 20000
    u = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    u[:12] = [0, 1, 2 ** 29 - 1, 2 ** 29, 2 ** 30 - 1, 2 ** 30, 2 ** 31 - 1, 2 ** 31, 3 * 2 ** 30, 2 ** 32 - 2 ** 29, 2 ** 32 - 2 ** 29 - 1, 2 ** 32 - 1]
    a = np.concatenate([10.0 ** rng.uniform(-12, 6, n - 4), [1.0, 0.25, 1.0 - 2.0 ** -30, 1.0 + 2.0 ** -29]])
    b = 10.0 ** rng.uniform(-20, 3, n)
    costh = np.concatenate([rng.uniform(-1, 1, n - 4), [1.0, -1.0, 1.0 - 1e-12, 0.0]])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:40, :2] = 0.0; d[:40, 2] = np.where(np.arange(40) % 2 == 0, 1.0, -1.0)   # along +-z: the untilted branch (DXY <= 1e-28)
    d = d.astype(np.float32)
    # compton_cdt1 is fed tau in (0, 1] and E in eV: a -> tau, b -> E / 1e5
    tau = rng.uniform(0.55, 1.0, n).astype(np.float32)
    e5 = rng.uniform(0.1, 1.25, n).astype(np.float32)
    with gpu_engine.create(case_dir("water"), device=0) as ctx:
        out = ctx.kat_fast64(u, a, b, costh, d)
        out_c = ctx.kat_fast64(u, tau.astype(np.float64), e5.astype(np.float64), costh, d)
    phi = 2.0 * np.pi * ((u.astype(np.float64) + 0.5) * 2.0 ** -32)
    # argument reduction is exact on the device (integer quarter turns); numpy's sin(2 pi x) carries the rounding of 2 pi x: 2e-16 x 6.3
    assert np.max(np.abs(out[:, 0] - np.sin(phi))) < 2e-15 and np.max(np.abs(out[:, 1] - np.cos(phi))) < 2e-15
    assert np.max(np.abs(out[:, 0] ** 2 + out[:, 1] ** 2 - 1.0)) < 5e-16
    assert np.max(np.abs(out[:, 2] * np.sqrt(a) - 1.0)) < 5e-16                           # 1 / sqrt(a)
    assert np.max(np.abs(out[:, 3] / np.sqrt(a / b) - 1.0)) < 1e-15                       # sqrt(a / b)
    ref = (1.0 - tau).astype(np.float64) / (tau.astype(np.float64) * (e5 * np.float32(1.0e5)).astype(np.float64) * 1.956951306108245e-6)
    ref = np.where(ref > 2.0, 1.99999999, ref)
    assert np.max(np.abs(out_c[:, 4] / ref - 1.0)) < 5e-16
    # rotate_double in numpy float64, statement for statement (the direction is float32, widened where the reference widens it)
    x, y, z = (d[:, k].copy() for k in range(3))
    dxy = (x * x + y * y).astype(np.float64)
    norm = dxy + (z * z).astype(np.float64)
    re = np.abs(norm - 1.0) > 1e-14
    s = np.where(re, 1.0 / np.sqrt(norm), 1.0)
    x, y, z = (np.where(re, (s * v.astype(np.float64)).astype(np.float32), v) for v in (x, y, z))
    dxy = np.where(re, (x * x + y * y).astype(np.float64), dxy)
    sp, cp = np.sin(phi), np.cos(phi)
    tilted = dxy > 1e-28
    X, Y, Z = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):  # the untilted lanes (dxy = 0) take the other branch below
        sdt = np.sqrt((1.0 - costh * costh) / dxy)
        ru = X * costh + sdt * ((x * z).astype(np.float64) * cp - Y * sp)
        rv = Y * costh + sdt * ((y * z).astype(np.float64) * cp + X * sp)
        rw = Z * costh - dxy * sdt * cp
    s0 = np.sqrt(1.0 - costh * costh)
    ru = np.where(tilted, ru, np.where(z > 0, s0 * cp, -s0 * cp))
    rv = np.where(tilted, rv, s0 * sp)
    rw = np.where(tilted, rw, np.where(z > 0, costh, -costh))
    want = np.stack([ru, rv, rw], axis=1).astype(np.float32).astype(np.float64)
    ulp = np.spacing(np.maximum(np.abs(want), 2.0 ** -20).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(out[:, 5:8] - want) <= ulp), float(np.max(np.abs(out[:, 5:8] - want) / ulp))
    assert np.max(np.abs(np.linalg.norm(out[:, 5:8], axis=1) - 1.0)) < 3e-7
    assert (~tilted).sum() >= 40
