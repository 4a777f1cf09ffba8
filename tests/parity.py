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


def poisson_z(img_a: np.ndarray, n_a: int, img_b: np.ndarray, n_b: int, min_counts: float = 30.0):
    """Per-pixel z-scores between two energy-weighted tallies (uint64, units of 0.01 eV) from n_a / n_b
    histories.  Variance model: compound Poisson, var(sum w) ~= sum w^2 ~= mean_w * sum w, with mean_w
    estimated per image class from the data (energy per detected photon ~ 6e6 units)."""
    a = img_a.astype(np.float64) / n_a
    b = img_b.astype(np.float64) / n_b
    w = 6.0e6  # typical tally weight (60 keV * 100); conservative: real weights are <= 1.25e7
    var = (img_a.astype(np.float64) * w * 1.6) / n_a ** 2 + (img_b.astype(np.float64) * w * 1.6) / n_b ** 2
    mask = (img_a.astype(np.float64) / w >= min_counts) & (img_b.astype(np.float64) / w >= min_counts)
    z = np.zeros_like(a)
    z[mask] = (a[mask] - b[mask]) / np.sqrt(var[mask])
    return z, mask
