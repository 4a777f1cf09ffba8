#!/usr/bin/env python3
"""Generate tests/golden/warp_kat.npz: known answers for the nearest-neighbour geometry warp (row f3).

Test infrastructure; runs in the development container (torch is installed here).  The reference warps a geometry with
vroc's SpatialTransformer (`MCGeometry.warp`, cbctmc/mc/geometry.py:386-439).  vroc is an un-vendored third-party
dependency of the reference (absent from /root/reference): its published algorithm (vroc/blocks.py, SpatialTransformer.
_warp_n_dim) is restated here with the SAME torch operations, so that the rounding of every step is torch's own:

    identity = meshgrid(arange(nx), arange(ny), arange(nz), indexing="ij")             # float32
    locs     = identity + displacement                                                   # voxels
    locs[i]  = 2 * (locs[i] / (shape[i] - 1) - 0.5)                                      # to [-1, 1]
    grid     = locs.permute(0, 2, 3, 4, 1)[..., [2, 1, 0]]                               # grid_sample wants (z, y, x) last-to-first
    warped   = F.grid_sample(image, grid, mode="nearest", align_corners=True)           # zero padding outside
    mask     = F.grid_sample(ones, grid, mode="nearest", align_corners=True) > 0         # default value where nothing was sampled
    warped[~mask] = float(default_value)

The cases hold ties (x + u exactly half-way between two voxels), samples within +-0.5 of both borders of every axis,
far-away samples, and volumes with even / odd / power-of-two / tiny extents.  tests/test_4d.py holds the HIP kernel
(csrc/warp.hip, through the C ABI) to these fixtures bit for bit.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"


def spatial_transform(image: torch.Tensor, displacement: torch.Tensor, default_value: float) -> torch.Tensor:
    """image [1, 1, x, y, z] float32, displacement [1, 3, x, y, z] float32 (voxels) -> warped image."""
    shape = image.shape[2:]
    identity = torch.stack(torch.meshgrid(*[torch.arange(0, s, dtype=torch.float32) for s in shape], indexing="ij"))[None]
    locs = identity + displacement
    for i in range(3):
        locs[:, i, ...] = 2 * (locs[:, i, ...] / (shape[i] - 1) - 0.5)
    grid = locs.permute(0, 2, 3, 4, 1)[..., [2, 1, 0]]
    warped = F.grid_sample(image, grid, align_corners=True, mode="nearest")
    mask = F.grid_sample(torch.ones_like(image), grid, align_corners=True, mode="nearest").to(torch.bool)
    warped[~mask] = float(default_value)
    return warped


def make_case(rng, shape, amplitude):
    nx, ny, nz = shape
    mats = rng.integers(1, 23, size=shape).astype(np.uint8)
    dens = rng.uniform(0.001, 2.7, size=shape).astype(np.float32)
    u = rng.uniform(-amplitude, amplitude, size=(3,) + shape).astype(np.float32)
    flat = [u[c].reshape(-1) for c in range(3)]
    coords = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")).reshape(3, -1).astype(np.float32)
    n = coords.shape[1]
    k = 0
    # exact ties: x + u = m + 0.5 for every representable m near the voxel, all axes
    for c, ext in enumerate(shape):
        for m in np.arange(-2, ext + 2):
            for idx in rng.integers(0, n, size=3):
                flat[c][idx] = np.float32(m + 0.5) - coords[c, idx]
                k += 1
    # just inside / outside the two borders of every axis (+- a few ulps around -0.5, 0, ext-1, ext-0.5)
    for c, ext in enumerate(shape):
        for target in (-0.5, 0.0, ext - 1.0, ext - 0.5):
            for ulps in (-3, -2, -1, 0, 1, 2, 3):
                t = np.float32(target)
                for _ in range(abs(ulps)):
                    t = np.nextafter(t, np.float32(np.inf if ulps > 0 else -np.inf), dtype=np.float32)
                idx = int(rng.integers(0, n))
                flat[c][idx] = t - coords[c, idx]
    # far away, both signs
    for c in range(3):
        for idx in rng.integers(0, n, size=4):
            flat[c][idx] = np.float32(rng.choice([-1e4, 1e4, -77.25, 300.5]))
    return mats, dens, u


def main():
    rng = np.random.default_rng(20261003)
    out = {}
    shapes = [((11, 9, 7), 4.0), ((16, 20, 24), 6.0), ((2, 3, 5), 2.0), ((33, 17, 8), 9.0), ((305, 12, 9), 15.0)]
    for k, (shape, amp) in enumerate(shapes):
        mats, dens, u = make_case(rng, shape, amp)
        disp = torch.from_numpy(u)[None]
        wm = spatial_transform(torch.from_numpy(mats.astype(np.float32))[None, None], disp, 1.0)[0, 0].numpy()
        wd = spatial_transform(torch.from_numpy(dens)[None, None], disp, np.float32(0.0013))[0, 0].numpy()
        assert np.array_equal(wm, np.round(wm)) and wm.min() >= 1
        out[f"materials_{k}"], out[f"densities_{k}"], out[f"field_{k}"] = mats, dens, u
        out[f"warped_materials_{k}"], out[f"warped_densities_{k}"] = wm.astype(np.uint8), wd.astype(np.float32)
        print(f"case {k}: shape {shape}, {int((wm != mats).sum())} voxels changed material, {int((wd == np.float32(0.0013)).sum())} took the default")
    out["n_cases"] = len(shapes)
    out["default_material"], out["default_density"] = 1, np.float32(0.0013)
    np.savez_compressed(GOLD / "warp_kat.npz", **out)
    print("torch", torch.__version__, "->", GOLD / "warp_kat.npz")


if __name__ == "__main__":
    main()
