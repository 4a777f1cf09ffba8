"""CPU tests: the restated oracle against the golden vectors captured from the real reference build.

ORACLE_MATH_LIBM must reproduce the reference's integer tallies exactly (same glibc as the container the
fixtures were generated in; on another libm a handful of histories may differ, which the test tolerates
only through an explicit, tiny budget).  ORACLE_MATH_PORTABLE must reproduce its own golden tallies exactly
on any machine -- it uses no libm.
"""
import ctypes as C
import re

import numpy as np
import pytest

import cases
import golden_util as gu
import oracle_lib as ol
import parity


def test_ranecu_known_answers():
    g = gu.load("rng_kat.npz")
    lib = ol.oracle()
    for row, f32, f64 in zip(g["init"], g["draws_f32"], g["draws_f64"]):
        batch, hpt, seed, s1, s2 = [int(v) for v in row]
        s = (C.c_int * 2)()
        lib.oracle_init_prng(batch, hpt, seed, s)
        assert (s[0], s[1]) == (s1, s2)
        got = np.array([lib.oracle_ranecu(s) for _ in range(len(f32))], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), f32.view(np.uint32))
        gotd = np.array([lib.oracle_ranecu_double(s) for _ in range(len(f64))])
        assert np.array_equal(gotd, f64)
    for m, a, s, want in g["abmodm"]:
        assert lib.oracle_abmodm(int(m), int(a), int(s)) == int(want)
    for b, h, s, want in g["update_seed"]:
        assert lib.oracle_update_seed(int(b), int(h), int(s)) == int(want)
    # values quoted in SURVEY.md 8c
    assert tuple(int(v) for v in g["init"][0][3:]) == (350588300, 585509557)


@pytest.fixture(scope="module")
def catphan_tables(case_dir, engine):
    ctx = engine.create(case_dir("catphan64"), device=-1)
    T = parity.tables_from_context(ctx)
    yield T
    ctx.close()


def test_physics_known_answers(catphan_tables):
    """GCOa, GRAa, rotate_double and source(): oracle (libm) == reference, call by call."""
    T = catphan_tables
    g = gu.load("physics_kat.npz")
    lib = ol.oracle()
    for mat, e, s1, s2, e_out, ct, t1, t2 in g["gcoa"]:
        s = (C.c_int * 2)(int(s1), int(s2))
        ef, c = C.c_float(e), C.c_double()
        lib.oracle_gcoa(C.byref(T.ct), C.byref(ef), C.byref(c), int(mat), s, ol.MATH_LIBM)
        assert (np.float32(ef.value), c.value, s[0], s[1]) == (np.float32(e_out), ct, int(t1), int(t2))
    for mat, e, idx, s1, s2, ct, t1, t2 in g["graa"]:
        s = (C.c_int * 2)(int(s1), int(s2))
        c = C.c_double()
        lib.oracle_graa(C.byref(T.ct), C.c_float(e), C.byref(c), int(mat), int(idx), s)
        assert (c.value, s[0], s[1]) == (ct, int(t1), int(t2))
    for u, v, w, costh, phi, u2, v2, w2 in g["rotate"]:
        d = (C.c_float * 3)(u, v, w)
        lib.oracle_rotate(d, costh, phi, ol.MATH_LIBM)
        assert (np.float32(d[0]), np.float32(d[1]), np.float32(d[2])) == (np.float32(u2), np.float32(v2), np.float32(w2))
        # portable trigonometry stays within float rounding of libm
        d = (C.c_float * 3)(u, v, w)
        lib.oracle_rotate(d, costh, phi, ol.MATH_PORTABLE)
        assert np.allclose([d[0], d[1], d[2]], [u2, v2, w2], atol=3e-7)
    for row in g["source"]:
        s = (C.c_int * 2)(int(row[0]), int(row[1]))
        pos, dr, en, av = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float(), C.c_int()
        lib.oracle_source(C.byref(T.ct), 0, s, pos, dr, C.byref(en), C.byref(av), ol.MATH_LIBM)
        got = [pos[0], pos[1], pos[2], dr[0], dr[1], dr[2], en.value, av.value, s[0], s[1]]
        assert np.array_equal(np.array(got, dtype=np.float64), row[2:])


def test_portable_math_accuracy():
    """The libm-free functions are accurate to ~1e-15, i.e. float results agree with libm except in rare ties."""
    lib = ol.oracle()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(1e-30, 1, 2000), rng.uniform(1, 1e6, 500)])
    got = np.array([lib.oracle_pm_log(v) for v in x])
    assert np.max(np.abs(got - np.log(x)) / np.maximum(np.abs(np.log(x)), 1e-300)) < 4e-15 or np.max(np.abs(got - np.log(x))) < 1e-15
    x = rng.uniform(-700, 5, 2000)
    got = np.array([lib.oracle_pm_exp(v) for v in x])
    assert np.max(np.abs(got / np.exp(x) - 1)) < 1e-14
    assert lib.oracle_pm_exp(-800.0) == 0.0
    x = rng.uniform(0, 2 * np.pi, 2000)
    for v in x:
        s, c = C.c_double(), C.c_double()
        lib.oracle_pm_sincos(v, C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(v)) < 4e-16 and abs(c.value - np.cos(v)) < 4e-16


def test_portable_expf_is_the_host_libm_expf():
    """gl_expf restates the C library's single-precision exp algorithm; on the build machine it equals libm's expf() on every
    float (oracle/check_libm.c, 4 278 190 082 inputs).  Here: every 4099th float plus the argument range the history loop uses,
    against the expf() of the machine the tests run on, and the table against its generator."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_exp2f_table", ol.ROOT / "oracle" / "gen_exp2f_table.py")
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    src = (ol.ROOT / "oracle" / "mcgpu_oracle.c").read_text()
    body = src[src.index("GL_EXP2_T[32] = {"):]
    body = body[:body.index("};")]
    assert [int(t, 16) for t in re.findall(r"0x[0-9a-f]{16}", body)] == gen.table()
    dev = (ol.ROOT / "4d-cbct-mc_amd" / "csrc" / "compat_math.inc").read_text()
    body = dev[dev.index("kExp2Table[32] = {"):]
    body = body[:body.index("};")]
    assert [int(t, 16) for t in re.findall(r"0x[0-9a-f]{16}", body)] == gen.table()
    # the table of pm_log (both copies) against ITS generator
    spec = importlib.util.spec_from_file_location("gen_log_table", ol.ROOT / "oracle" / "gen_log_table.py")
    genl = importlib.util.module_from_spec(spec); spec.loader.exec_module(genl)
    want = [v for pair in genl.table() for v in pair]
    for text, name in ((src, "PM_LOG_T[64][2] = {"), (dev, "kLogTable[64][2] = {")):
        body = text[text.index(name):]
        body = body[:body.index("};")]
        got = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", body)]
        assert got == want
    lib = ol.oracle()
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    bits = np.concatenate([np.arange(0, 2**32, 4099 * 64, dtype=np.uint64).astype(np.uint32),
                           np.random.default_rng(5).uniform(-104.5, 0.5, 40000).astype(np.float32).view(np.uint32)])
    x = bits.view(np.float32)
    x = x[~np.isnan(x)]
    got = np.array([lib.oracle_gl_expf(float(v)) for v in x], dtype=np.float32)
    want = np.array([libm.expf(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_reproduces_reference_tallies(name, case_dir, engine):
    g = gu.load(f"case_{name}.npz")
    nb, hpt = [int(v) for v in g["nbatch_hpt"]]
    with engine.create(case_dir(name), device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        size = T.image_size()
        for p in range(ctx.num_projections):
            seed = 42 + 1000 * p
            want = gu.dense(g, "ref", p, size)
            got, cnt = T.track(p, seed, 0, nb, hpt, ol.MATH_LIBM, n_threads=1)
            ndiff = np.count_nonzero(got != want)
            # bit-exact on the glibc the fixtures were made with; a foreign libm may flip a few histories
            assert ndiff <= (0 if ol.reference_available() else 40), f"{name} p{p}: {ndiff} tally words differ from the reference"
            got_mt, _ = T.track(p, seed, 0, nb, hpt, ol.MATH_LIBM, n_threads=4)
            assert np.array_equal(got, got_mt), "OpenMP batches must give the same integer tallies"
            want_pm = gu.dense(g, "portable", p, size)
            got_pm, _ = T.track(p, seed, 0, nb, hpt, ol.MATH_PORTABLE, n_threads=4)
            assert np.array_equal(got_pm, want_pm), f"{name} p{p}: portable-math tallies changed"
            # the two math modes describe the same physics: nearly all histories identical
            assert np.count_nonzero(want_pm != want) <= max(8, want.astype(bool).sum() // 500)
        assert cnt.histories == nb * hpt


def test_oracle_against_live_reference(case_dir):
    """When oracle/_ref is present: longer live run against the reference's own code, incl. batch offsets."""
    if not ol.reference_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    import os
    import sys
    ref = ol.Reference()
    sys.stdout.flush()
    saved, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        ref.load(case_dir("catphan64"))
    finally:
        os.dup2(saved, 1)
        os.close(null)
        os.close(saved)
    T = ref.tables()
    want = ref.track(0, 1234, 37, 600, 77)
    got, _ = T.track(0, 1234, 37, 600, 77, ol.MATH_LIBM, n_threads=4)
    assert np.array_equal(got, want)
