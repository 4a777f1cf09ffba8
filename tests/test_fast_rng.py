"""The FAST personality's random-number streams (track_common.inc: rng_init_history / rng_u32 / rng_f), pinned and tested directly.

* CPU: the numpy restatement oracle/fast_rng.py is held to the published known-answer vectors of Philox4x32 (Random123
  distribution, `kat_vectors`: 10 and 7 rounds) and to a plain-Python-integer version of itself; properties of the
  multiply-with-carry step (period arithmetic, the two fixed points, the seeding's range).
* GPU: the device streams equal the restatement word for word for history ids {0, 1, 2^32 - 1, 2^32, 2^40, ...} x projections
  {0, 893} (an independent pin: `fast_pin.json` only pins the kernel against itself), and the first 32 deviates of 2^24 consecutive
  history ids pass chi^2 tests in one, two and three dimensions and show no correlation between neighbouring ids or projections
  -- with the same statistics computed for a Philox4x32-10-per-draw generator as the yardstick.
"""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import fast_rng as fr  # noqa: E402


# ------------------------------------------------------------------------------------------------ CPU
PHILOX_KAT = [  # rounds, counter, key, expected   (Random123 kat_vectors)
    (10, [0, 0, 0, 0], [0, 0], "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
    (10, [0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
    (10, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], "d16cfe09 94fdcceb 5001e420 24126ea1"),
    (7, [0, 0, 0, 0], [0, 0], "5f6fb709 0d893f64 4f121f81 4f730a48"),
    (7, [0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, "5207ddc2 45165e59 4d8ee751 8c52f662"),
    (7, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], "4dfccaba 190a87f0 c47362ba b6b5242a"),
]


@pytest.mark.parametrize("rounds,ctr,key,expected", PHILOX_KAT)
def test_philox_restatement_matches_published_vectors(rounds, ctr, key, expected):
    out = fr.philox4x32([np.array([c], dtype=np.uint64) for c in ctr], key, rounds)
    assert " ".join("%08x" % int(v[0]) for v in out) == expected


KAT_IDS = [0, 1, 2, 255, 256, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 40, 2 ** 40 + 12345, 99_999_999, 2 ** 63 + 7]


def test_numpy_restatement_equals_plain_python_integers():
    for proj in (0, 893):
        got = fr.streams_u32(KAT_IDS, seed=271828, projection=proj, n_draws=40)
        for row, hist in zip(got, KAT_IDS):
            assert [int(v) for v in row] == fr.streams_python(hist, 271828, proj, 40)


def test_mwc_parameters_and_seeding_range():
    a = fr.MWC_A
    # period of a lag-1 MWC with base b = 2^32: the order of b modulo m = a b - 1.  m and (m - 1) / 2 = a 2^31 - 1 prime ("safe
    # prime") => the order is (m - 1) / 2 or m - 1, i.e. >= a 2^31 - 1 ~ 2^62.99.  Primality by deterministic Miller-Rabin.
    def is_prime(n):
        d, s = n - 1, 0
        while d % 2 == 0:
            d //= 2; s += 1
        for w in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):  # deterministic for n < 3.3e24
            if w % n == 0:
                continue
            x = pow(w, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True
    m = a * 2 ** 32 - 1
    assert is_prime(m) and is_prime((m - 1) // 2)
    assert pow(2 ** 32, (m - 1) // 2, m) == 1  # b is a quadratic residue: the period is exactly a 2^31 - 1
    # fixed points of the step: (0, 0) and (2^32 - 1, a - 1)
    assert fr.mwc_step(np.uint64(0), np.uint64(0)) == (0, 0)
    x, c = fr.mwc_step(np.uint64(2 ** 32 - 1), np.uint64(a - 1))
    assert (int(x), int(c)) == (2 ** 32 - 1, a - 1)
    # seeding: c = umulhi(w, a - 1) + 1 lies in [1, a - 1]; a - 1 only for w = 2^32 - 1, and then the fixed point is avoided explicitly
    assert ((2 ** 32 - 1) * (a - 1) >> 32) + 1 == a - 1 and ((2 ** 32 - 2) * (a - 1) >> 32) + 1 == a - 2
    ids = np.arange(1 << 16, dtype=np.uint64)
    x0, c0 = fr.seed_streams(ids, 42, 7)
    assert c0.min() >= 1 and c0.max() <= a - 1 and not np.any((x0 == 2 ** 32 - 1) & (c0 == a - 1))


def test_battery_generators_equal_the_numpy_restatement():
    """oracle/rng_battery.c (the TestU01-style battery whose verdict at 2^36 bytes is profiles/r06e_*) restates both generators in C:
    its known answers -- first six outputs of the streams of four history ids, seed 42, projection 893 -- equal oracle/fast_rng.py's
    (product: Philox4x32-7 seeding + multiply-with-carry; yardstick: Philox4x32-10 per draw, counter {id, projection, k / 4})."""
    import subprocess
    exe = ROOT / "oracle" / "rng_battery"
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "rng_battery"], check=True, capture_output=True, timeout=300)
    lines = subprocess.run([str(exe), "--kat"], capture_output=True, text=True, check=True, timeout=60).stdout.strip().split("\n")
    assert len(lines) == 8
    for line in lines:
        head, words = line.split(":")
        t = head.split()
        gen, hist = int(t[2]), int(t[4])
        got = [int(w, 16) for w in words.split()]
        if gen == 0:
            want = fr.streams_u32(np.array([hist], dtype=np.uint64), 42, 893, 6)[0].tolist()
        else:
            want = []
            for k4 in range(0, 8, 4):
                want += [int(v) for v in fr.philox4x32([hist & fr.MASK, hist >> 32, 893, k4 >> 2], [42, fr.KEY1], rounds=10)]
            want = want[:6]
        assert got == want, line


def test_battery_rejects_a_careless_seeding():
    """`rng_battery --control`: the same multiply-with-carry step seeded straight from the history id (no Philox hash) must FAIL the
    battery -- evidence that its draw-major tests see what a per-history seeding can get wrong.  (The product's verdict at 2^37.65
    bytes -- 128 statistics, none suspect -- is profiles/r06e_rng_battery_2p37_bytes.txt; that run takes 25 minutes of 8 cores.)"""
    import subprocess
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "rng_battery"], check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(ROOT / "oracle" / "rng_battery"), "--control"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:]
    fails = [l for l in r.stdout.split("\n") if l.rstrip().endswith("FAIL")]
    assert len(fails) >= 16 and any("collisions" in l for l in fails) and any("birthday" in l for l in fails) and any("matrix rank" in l for l in fails)
    record = (ROOT / "profiles" / "r06e_rng_battery_2p37_bytes.txt").read_text()
    assert "128 statistics: 0 suspect" in record and "0 FAIL" in record and record.count("PRODUCT:") == 4 and record.count("(yardstick) |") == 4


def test_deviate_is_never_zero_or_one():
    f = fr.to_float(np.array([0, 255, 256, 2 ** 32 - 1], dtype=np.uint32))
    assert f[0] == np.float32(2.0 ** -26) and f[1] == f[0] and f[2] == np.float32(2.0 ** -24 + 2.0 ** -26)
    assert f[3] < 1.0 and f[3] == np.float32(1.0 - 2.0 ** -24)  # (2^24 - 1) 2^-24 + 2^-26 rounds to 1 - 2^-24


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx(engine, case_dir):
    with engine.create(case_dir("air"), device=0) as c:
        yield c


@pytest.mark.gpu
def test_device_streams_equal_the_restatement(ctx):
    for proj in (0, 893):
        for seed in (271828, 1):
            want = fr.streams_u32(KAT_IDS, seed=seed, projection=proj, n_draws=64)
            got = ctx.kat_rng_streams(seed, proj, 64, ids=KAT_IDS)
            assert np.array_equal(got, want)
    # consecutive ids through the first_id route, across the 2^32 boundary
    first = 2 ** 32 - 1000
    got = ctx.kat_rng_streams(42, 5, 16, first_id=first, n_ids=2000)
    assert np.array_equal(got, fr.streams_u32(np.arange(first, first + 2000, dtype=np.uint64), 42, 5, 16))
    # and the float the kernel draws from it (mcgpu_kat_rng: rng_f of history `batch`, projection 0)
    f = ctx.kat_rng("fast", 271828, 12345, 0, 64)
    assert np.array_equal(f, fr.to_float(fr.streams_u32([12345], 271828, 0, 64)[0]))


def _chi2_z(counts, expected):
    """chi^2 of a histogram against a flat expectation as a z-score (Wilson-Hilferty would be overkill at >= 255 cells)."""
    chi2 = float((((counts - expected) ** 2) / expected).sum())
    dof = counts.size - 1
    return (chi2 - dof) / math.sqrt(2.0 * dof)


def _stream_statistics(ctx, generator, n_ids_log2=24, n_draws=32, seed=20240607, proj=447):
    """Accumulates, over 2^n_ids_log2 consecutive history ids in chunks: 1-D histograms of the 24-bit deviate (4096 cells) per draw
    position, serial pairs (64 x 64) and triples (16^3) of consecutive draws of a stream, and the correlation of draw k between ids
    n and n + 1 and between projections p and p + 1."""
    chunk = 1 << 20
    n_chunks = (1 << n_ids_log2) // chunk
    h1 = np.zeros((n_draws, 4096), dtype=np.int64)
    h2 = np.zeros(64 * 64, dtype=np.int64)
    h3 = np.zeros(16 ** 3, dtype=np.int64)
    h2_ids = np.zeros(64 * 64, dtype=np.int64)    # (draw k of id n, draw k of id n + 1)
    h2_proj = np.zeros(64 * 64, dtype=np.int64)   # (draw k of projection p, draw k of projection p + 1)
    corr_ids = np.zeros(n_draws); corr_proj = np.zeros(n_draws); corr_lag1 = 0.0
    n_pairs_ids = 0
    for ch in range(n_chunks):
        u = ctx.kat_rng_streams(seed, proj, n_draws, first_id=ch * chunk, n_ids=chunk, generator=generator) >> np.uint32(8)   # 24 bits
        v = ctx.kat_rng_streams(seed, proj + 1, n_draws, first_id=ch * chunk, n_ids=chunk, generator=generator) >> np.uint32(8)
        for k in range(n_draws):
            h1[k] += np.bincount(u[:, k] >> np.uint32(12), minlength=4096)
        a6 = (u >> np.uint32(18)).astype(np.int64)
        h2 += np.bincount((a6[:, :-1] * 64 + a6[:, 1:]).ravel(), minlength=4096)
        a4 = (u >> np.uint32(20)).astype(np.int64)
        h3 += np.bincount((a4[:, :-2] * 256 + a4[:, 1:-1] * 16 + a4[:, 2:]).ravel(), minlength=4096)
        h2_ids += np.bincount((a6[:-1] * 64 + a6[1:]).ravel(), minlength=4096)
        b6 = (v >> np.uint32(18)).astype(np.int64)
        h2_proj += np.bincount((a6 * 64 + b6).ravel(), minlength=4096)
        x = u.astype(np.float64) * 2.0 ** -24 - 0.5 + 2.0 ** -25
        y = v.astype(np.float64) * 2.0 ** -24 - 0.5 + 2.0 ** -25
        corr_ids += (x[:-1] * x[1:]).sum(axis=0)
        corr_proj += (x * y).sum(axis=0)
        corr_lag1 += float((x[:, :-1] * x[:, 1:]).sum())
        n_pairs_ids += chunk - 1
    n = n_chunks * chunk
    z = {
        "chi2_1d_worst_draw": max(abs(_chi2_z(h1[k], n / 4096)) for k in range(n_draws)),
        "chi2_1d_all_draws": _chi2_z(h1.sum(axis=0), n * n_draws / 4096),
        "chi2_pairs": _chi2_z(h2, n * (n_draws - 1) / 4096),
        "chi2_triples": _chi2_z(h3, n * (n_draws - 2) / 4096),
        "chi2_neighbouring_ids": _chi2_z(h2_ids, n_pairs_ids * n_draws / 4096),
        "chi2_neighbouring_projections": _chi2_z(h2_proj, n * n_draws / 4096),
        # sum of products of two independent centred uniforms: variance 1/144 per term
        "corr_ids_worst_draw": float(np.abs(corr_ids).max() / math.sqrt(n_pairs_ids / 144.0)),
        "corr_projections_worst_draw": float(np.abs(corr_proj).max() / math.sqrt(n / 144.0)),
        "corr_lag1": abs(corr_lag1) / math.sqrt(n * (n_draws - 1) / 144.0),
    }
    return z


@pytest.mark.gpu
def test_stream_statistics_against_philox_per_draw_yardstick(ctx):
    """2^22 consecutive history ids (MCGPU_RNG_TEST_LOG2=24: 2^24, the run recorded under profiles/) x 32 deviates (a history of
    the bench workloads consumes ~16-40).  Every statistic is a z-score; the production generator must stay below 4.5 (the worst
    of ~32 looks: 4 sigma on one of them happens by chance once in 500 runs) wherever the yardstick does, and a yardstick failure
    would mean the test itself is broken."""
    import os
    log2 = int(os.environ.get("MCGPU_RNG_TEST_LOG2", "22"))
    prod = _stream_statistics(ctx, generator=0, n_ids_log2=log2)
    yard = _stream_statistics(ctx, generator=1, n_ids_log2=log2)
    print("history ids: 2^%d x 32 deviates" % log2)
    print("production (Philox4x32-7 -> MWC):", {k: round(v, 2) for k, v in prod.items()})
    print("yardstick  (Philox4x32-10 per draw):", {k: round(v, 2) for k, v in yard.items()})
    for k, v in yard.items():
        assert abs(v) < 4.5, ("yardstick", k, v)
    for k, v in prod.items():
        assert abs(v) < 4.5, ("production", k, v)
