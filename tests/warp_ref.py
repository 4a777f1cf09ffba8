"""numpy restatement of the reference's geometry warp (test infrastructure): vroc SpatialTransformer = identity + field,
normalise, `grid_sample(mode="nearest", align_corners=True)`, default outside -- every step in float32 like torch.
Pinned against torch itself by tests/golden/warp_kat.npz (oracle/gen_warp_golden.py; test_4d.py::test_warp_restatement_...)."""
import numpy as np


def warp_nearest(mats, dens, field, default_material, default_density):
    """mats/dens [x, y, z], field [3, x, y, z] in voxels -> (warped materials, warped densities)."""
    shape = mats.shape
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")).astype(np.float32)
    inside = np.ones(shape, dtype=bool)
    src = []
    for c, n in enumerate(shape):
        loc = (idx[c] + field[c].astype(np.float32)).astype(np.float32)
        t = (loc / np.float32(n - 1)).astype(np.float32)
        t = (np.float32(2) * (t - np.float32(0.5)).astype(np.float32)).astype(np.float32)
        s = np.rint((((t + np.float32(1)).astype(np.float32) / np.float32(2)).astype(np.float32) * np.float32(n - 1)).astype(np.float32))
        inside &= (s >= 0) & (s <= n - 1)
        src.append(np.clip(s, 0, n - 1).astype(np.int64))
    wm = np.where(inside, mats[src[0], src[1], src[2]], default_material).astype(mats.dtype)
    wd = np.where(inside, dens[src[0], src[1], src[2]], np.float32(default_density)).astype(np.float32)
    return wm, wd
