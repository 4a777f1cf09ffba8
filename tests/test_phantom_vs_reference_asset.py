"""Development-container check: the procedural Catphan604 generator reproduces the phantom bundled with the
reference (cbctmc/assets/geometries/catphan604_geometry.pkl.gz) voxel for voxel.  Skipped where the reference
tree is absent (GPU box)."""
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
