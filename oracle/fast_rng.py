"""TEST INFRASTRUCTURE (never imported by the product): an independent numpy restatement of the FAST personality's random-number
streams, 4d-cbct-mc_amd/csrc/track_common.inc `rng_init_history` / `rng_u32` / `rng_f`.

The reference has no counterpart (its generator is RANECU, K.cu:841-894, which the COMPAT personality reproduces bit for bit);
north_star names "XORWOW/Philox per lane".  The FAST kernel gives every HISTORY its own stream:

  Philox4x32-7( counter = {id_lo, id_hi, projection, 0x4d43475}, key = {seed, 0xCB435443} )          (Salmon et al., SC11)
      -> x = out0 ^ out2,  c = umulhi(out1 ^ out3, a - 1) + 1     (1 <= c <= a - 1; the fixed point (2^32 - 1, a - 1) is moved to c = a - 2)
  multiply-with-carry, base 2^32, lag 1, a = 4294584393:  t = a x + c;  x' = t mod 2^32;  c' = t div 2^32;  output x'
  deviate = (x' >> 8) 2^-24 + 2^-26   (never 0, never 1)

`philox4x32` is pinned by the known-answer vectors of the Random123 distribution (tests/test_fast_rng.py), so the restatement
does not lean on the kernel it checks.
"""
from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57          # Philox4x32 multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85          # Weyl key increments
MWC_A = 4294584393                        # MWC multiplier (a 2^32 - 1 and a 2^31 - 1 prime)
KEY1, CTR3 = 0xCB435443, 0x4D43475       # the kernel's fixed key / counter words
MASK = 0xFFFFFFFF


def philox4x32(ctr, key, rounds: int = 10):
    """ctr: 4 arrays (or ints) of uint32 words, key: 2.  Returns the 4 output words as uint64 arrays holding 32-bit values."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in ctr]
    k = [np.uint64(int(v) & MASK) if np.isscalar(v) else (np.asarray(v, dtype=np.uint64) & MASK) for v in key]
    for _ in range(rounds):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + np.uint64(W0)) & MASK, (k[1] + np.uint64(W1)) & MASK]
    return c


def seed_streams(ids, seed: int, projection: int):
    """(x, c) of the lane generator for history ids `ids` (uint64) of one projection."""
    ids = np.asarray(ids, dtype=np.uint64)
    o = philox4x32([ids & MASK, ids >> np.uint64(32), np.full(ids.shape, projection, np.uint64), np.full(ids.shape, CTR3, np.uint64)],
                   [seed, KEY1], rounds=7)
    x = o[0] ^ o[2]
    c = (((o[1] ^ o[3]) * np.uint64(MWC_A - 1)) >> np.uint64(32)) + np.uint64(1)
    # (2^32 - 1, a - 1) is a fixed point of the step; the seeding can produce c = a - 1 (only from the word 2^32 - 1): moved off it
    c = np.where((x == np.uint64(MASK)) & (c == np.uint64(MWC_A - 1)), c - np.uint64(1), c)
    return x, c


def mwc_step(x, c):
    """One step; returns (x', c').  a x + c < 2^64 always (a < 2^32, c < a)."""
    t = np.uint64(MWC_A) * x + c
    return t & MASK, t >> np.uint64(32)


def streams_u32(ids, seed: int, projection: int, n_draws: int) -> np.ndarray:
    """uint32[len(ids), n_draws]: the raw outputs of each history's stream."""
    x, c = seed_streams(ids, seed, projection)
    out = np.empty((x.size, n_draws), dtype=np.uint32)
    for k in range(n_draws):
        x, c = mwc_step(x, c)
        out[:, k] = x.astype(np.uint32)
    return out


def to_float(u32) -> np.ndarray:
    """rng_f: the 24 upper bits as k 2^-24 + 2^-26 in float32 (one fused multiply-add, exact in float32 for every k)."""
    k = (np.asarray(u32, dtype=np.uint32) >> np.uint32(8)).astype(np.float64)
    return (k * 2.0 ** -24 + 2.0 ** -26).astype(np.float32)


def streams_python(hist: int, seed: int, projection: int, n_draws: int):
    """The same in plain Python integers (cross-check of the numpy version on a few ids)."""
    c = [hist & MASK, (hist >> 32) & MASK, projection & MASK, CTR3]
    k0, k1 = seed & MASK, KEY1
    for _ in range(7):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & MASK, (p0 >> 32) ^ c[3] ^ k1, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    x, cc = c[0] ^ c[2], (((c[1] ^ c[3]) * (MWC_A - 1)) >> 32) + 1
    if x == MASK and cc == MWC_A - 1:
        cc -= 1
    out = []
    for _ in range(n_draws):
        t = MWC_A * x + cc
        x, cc = t & MASK, t >> 32
        out.append(x)
    return out
