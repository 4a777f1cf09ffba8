"""Phantom geometries against the reference's bundled assets.  Development-container checks (skipped where the reference
tree is absent, i.e. on the GPU box): the procedural Catphan604 generator reproduces
cbctmc/assets/geometries/catphan604_geometry.pkl.gz voxel for voxel, and the shipped CIRS base geometry
(assets/geometries/base_cirs_geometry.npz) equals base_cirs_geometry.pkl.gz.  Everywhere: the shipped CIRS asset is
pinned by its SHA-256, and the insert builders are checked against independent restatements of geo.py:749-878."""
import hashlib
import gzip
import pickle
from pathlib import Path

import numpy as np
import pytest

import cases

ASSET = Path("/root/reference/cbctmc/assets/geometries/catphan604_geometry.pkl.gz")


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("cbctmc") or module.startswith("ipmi") or module.startswith("vroc"):
            return _Stub
        return super().find_class(module, name)


@pytest.mark.skipif(not ASSET.exists(), reason="reference assets not present")
def test_catphan_generator_matches_bundled_asset():
    with gzip.open(ASSET, "rb") as f:
        ref = _Unpickler(f).load()
    mine = cases.geometry.MCCatPhan604Geometry()
    assert ref.materials.shape == (500, 500, 500)
    assert np.array_equal(np.asarray(ref.materials), mine.materials)
    assert np.array_equal(np.asarray(ref.densities, dtype=np.float32), mine.densities)


CIRS_ASSET = Path("/root/reference/cbctmc/assets/geometries/base_cirs_geometry.pkl.gz")


@pytest.mark.skipif(not CIRS_ASSET.exists(), reason="reference assets not present")
def test_shipped_cirs_base_geometry_equals_bundled_asset():
    with gzip.open(CIRS_ASSET, "rb") as f:
        ref = _Unpickler(f).load()
    mine = cases.geometry.MCCIRSPhantomGeometry.from_base_geometry()
    assert mine.image_shape == (305, 300, 152) and mine.image_spacing == tuple(ref.image_spacing)
    assert np.array_equal(np.asarray(ref.materials), mine.materials)
    assert np.array_equal(np.asarray(ref.densities, dtype=np.float32), mine.densities)


def test_shipped_cirs_base_geometry_is_pinned():
    g = cases.geometry.MCCIRSPhantomGeometry.from_base_geometry()
    assert hashlib.sha256(g.materials.tobytes()).hexdigest().startswith("c8bf96c16c10209e")
    assert hashlib.sha256(g.densities.tobytes()).hexdigest().startswith("f7b4e8a84a6f1257")
    M = cases.materials
    pairs = {(int(m), round(float(d), 4)) for m, d in zip(*np.unique(np.stack([g.materials.ravel().astype(np.float64), g.densities.ravel()]), axis=1))}
    assert pairs == {(M.material_number("air"), 0.0013), (M.material_number("h2o"), 0.207), (M.material_number("soft_tissue"), 1.0),
                     (M.material_number("red_marrow"), 1.03), (M.material_number("bone_020"), 1.14), (M.material_number("bone_050"), 1.4),
                     (M.material_number("bone_100"), 1.92)}


def test_cirs_inserts_follow_the_reference_recipes():
    """Brute-force restatements on meshgrids (the reference's own formulation, geo.py:749-878) of the masks the
    bounding-box-free broadcasting code in geometry.py builds."""
    G, M = cases.geometry.MCCIRSPhantomGeometry, cases.materials
    shape, c = (60, 50, 48), np.array([31, 24, 20])
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    sphere = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 <= 15.0 ** 2
    assert np.array_equal(G.create_spherical_mask(15.0, shape, c), sphere)
    cz = c[2] + 7.5
    bore = ((x - c[0]) ** 2 + (y - c[1]) ** 2 <= 1.5 ** 2) & (z >= cz - 7.5) & (z <= cz + 7.5)
    want = sphere & ~bore
    assert np.array_equal(G.create_cirs_insert(shape, c), want) and bore.sum() > 0
    base = G.from_base_geometry()
    g = base.place_insert(shift=(0, 0, 5))
    changed = g.materials != base.materials
    assert 13000 < changed.sum() < 14200  # a 30 mm sphere (14137 voxels) minus the bore, minus voxels that already were soft tissue
    assert set(np.unique(g.materials[changed])) == {M.material_number("soft_tissue")}
    zs = np.nonzero(changed.any(axis=(0, 1)))[0]
    assert zs.min() == 71 + 5 - 15 and zs.max() == 71 + 5 + 14  # the upper pole voxel lies in the bore
    lp = base.place_line_pair_insert(gap=4)
    assert lp.image_shape == (1220, 300, 152) and lp.image_spacing == (0.25, 1.0, 1.0)
    al = lp.materials == M.material_number("aluminium")
    assert al.sum() == 4 * 16 * 40 * 40
    xs = np.nonzero(al.any(axis=(1, 2)))[0]
    assert xs.min() == 238 * 4 - 2 * 32 and xs.max() == 238 * 4 - 2 * 32 + 3 * 32 + 15
    lung_eq = (lp.materials == M.material_number("h2o")) & np.isclose(lp.densities, 0.207)
    assert lung_eq[xs.min() + 16: xs.min() + 32, 141, 71].all()
    assert np.array_equal(lp.materials[::4][:222], base.materials[:222])  # untouched in front of the insert
