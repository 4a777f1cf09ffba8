"""Helpers shared by the parity tests: oracle tables from the product's own host model."""
from __future__ import annotations

import numpy as np

import oracle_lib as ol


def tables_from_context(ctx) -> ol.TableSet:
    """Feed the CPU oracle with the host tables the product parsed (reference layouts via the C ABI).
    The tables themselves are pinned against the reference by tests/test_host_tables.py."""
    a = {
        "voxel_mat_dens": ctx.host_table("voxel_mat_dens", "<f4"),
        "mfp_woodcock": ctx.host_table("mfp_woodcock", "<f4"),
        "mfp_a": ctx.host_table("mfp_a", "<f4"), "mfp_b": ctx.host_table("mfp_b", "<f4"),
        "xco": ctx.host_table("xco", "<f4"), "pco": ctx.host_table("pco", "<f4"),
        "aco": ctx.host_table("aco", "<f4"), "bco": ctx.host_table("bco", "<f4"),
        "pmax": ctx.host_table("pmax", "<f4"), "itlco": ctx.host_table("itlco"), "ituco": ctx.host_table("ituco"),
        "fco": ctx.host_table("fco", "<f4"), "uico": ctx.host_table("uico", "<f4"), "fj0": ctx.host_table("fj0", "<f4"),
        "noscco": ctx.host_table("noscco", "<i4"), "espc": ctx.host_table("espc", "<f4"),
        "espc_cutoff": ctx.host_table("espc_cutoff", "<f4"), "espc_alias": ctx.host_table("espc_alias", "<i2"),
        "source_data": ctx.host_table("source_data"), "detector_data": ctx.host_table("detector_data"),
    }
    nvox = (ctx.geti("num_voxels_x"), ctx.geti("num_voxels_y"), ctx.geti("num_voxels_z"))
    return ol.TableSet(a, nvox, ctx.host_table("inv_voxel_size", "<f4"), ctx.host_table("size_bbox", "<f4"),
                       ctx.geti("num_energy_values"), np.float32(ctx.getf("e0")), np.float32(ctx.getf("ide")),
                       ctx.geti("num_spectrum_bins"))


def measured_z(img_a: np.ndarray, n_a: int, img_b: np.ndarray, w2_b: np.ndarray, n_b: int, min_hits: float = 30.0, w2_a=None):
    """Per-word z-scores between two energy-weighted tallies (units of 0.01 eV) of n_a / n_b histories, with MEASURED
    variances.  A tally word is a sum over histories of a weight w_i (0 for most): its variance per history is
    E[w^2] - E[w]^2.  `w2_b` is the sum of squared weights tallied beside `img_b` (oracle_lib.track_with_variance); sample a
    (the GPU's, which tallies no squares) uses `w2_a` if given.

    Under the hypothesis being tested (same distribution) the best estimate of a word's rate is the POOLED one,
    m = (a + b) / (n_a + n_b), and E[w^2] = m * rho with rho = sum w^2 / sum w the measured mean-square-to-mean weight of
    that word: var(a/n_a - b/n_b) = (m rho - m^2) (1/n_a + 1/n_b).  Pooling matters: a variance taken from sample b alone
    is correlated with b's own fluctuation and biases the mean z by ~1/(2 sqrt(hits)), and a mask on b's hits selects the
    words where b fluctuated up (seen as a -0.09 sigma mean offset over 1.2e4 blocks before this form was used).
    Words expected to hold fewer than `min_hits` effective hits (m n_b / rho) in sample b are masked out.
    Arrays may be block sums (both sums are additive)."""
    a, b, q = img_a.astype(np.float64), img_b.astype(np.float64), np.asarray(w2_b, dtype=np.float64)
    if w2_a is not None:
        q = q + np.asarray(w2_a, dtype=np.float64)
        base = a + b
    else:
        base = b
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = np.where(base > 0, q / base, 0.0)
        m = (a + b) / float(n_a + n_b)
        hits = np.where(rho > 0, m * n_b / rho, 0.0)
    var = np.maximum(m * rho - m * m, 0.0) * (1.0 / n_a + 1.0 / n_b)
    mask = (hits >= min_hits) & (var > 0)
    z = np.zeros_like(a)
    z[mask] = (a[mask] / n_a - b[mask] / n_b) / np.sqrt(var[mask])
    return z, mask


def blocks(img: np.ndarray, k: int = 3) -> np.ndarray:
    """Sum [4, nz, nx] tallies over k x k pixel blocks (ragged edges dropped)."""
    c, nz, nx = img.shape
    return img[:, : nz // k * k, : nx // k * k].reshape(c, nz // k, k, nx // k, k).sum(axis=(2, 4))


def class_energy_z(img_a, n_a, img_b, w2_b, n_b):
    """z of the detected energy per history of each scatter class (whole-image sums), measured variances as above;
    NaN for a class the reference sample holds fewer than 200 effective hits of."""
    out = []
    for k in range(img_a.shape[0]):
        z, m = measured_z(np.array([img_a[k].sum(dtype=np.float64)]), n_a, np.array([img_b[k].sum(dtype=np.float64)]),
                          np.array([np.asarray(w2_b[k], dtype=np.float64).sum()]), n_b, min_hits=200.0)
        out.append(float(z[0]) if m[0] else float("nan"))
    return out
