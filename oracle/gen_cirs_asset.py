#!/usr/bin/env python3
"""Convert the reference's bundled CIRS phantom geometry into the data asset this package ships.

Development-container tool (test infrastructure; needs /root/reference).  The reference loads
`cbctmc/assets/geometries/base_cirs_geometry.pkl.gz` in `MCCIRSPhantomGeometry.from_base_geometry`
(cbctmc/mc/geometry.py:642-649): a pickled `MCGeometry` (a segmented CT of the physical phantom, so there is
no recipe to regenerate it from).  A pickle of a foreign class cannot be loaded without that class; the arrays
are re-stored as a plain `.npz` (materials uint8, densities float32, spacing/origin/direction), which
`geometry.MCCIRSPhantomGeometry.from_base_geometry()` reads.  Data only: no code travels.

Usage: python oracle/gen_cirs_asset.py
"""
from __future__ import annotations

import gzip
import hashlib
import pickle
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
SRC = Path("/root/reference/cbctmc/assets/geometries/base_cirs_geometry.pkl.gz")
DST = ROOT / "4d-cbct-mc_amd" / "assets" / "geometries" / "base_cirs_geometry.npz"


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


class StubUnpickler(pickle.Unpickler):
    """Resolves every non-numpy class to an attribute bag (the reference package is not importable here)."""

    def find_class(self, module, name):
        if module.split(".")[0] in ("cbctmc", "ipmi", "vroc"):
            return _Stub
        return super().find_class(module, name)


def load_reference_geometry(path=SRC):
    with gzip.open(path, "rb") as f:
        return StubUnpickler(f).load()


def main():
    g = load_reference_geometry()
    mats = np.ascontiguousarray(g.materials, dtype=np.uint8)
    dens = np.ascontiguousarray(g.densities, dtype=np.float32)
    assert g.mus is None
    DST.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(DST, materials=mats, densities=dens, image_spacing=np.array(g.image_spacing, dtype=np.float64),
                        image_origin=np.array(g.image_origin, dtype=np.float64),
                        image_direction=np.array(g.image_direction, dtype=np.float64))
    print(DST, DST.stat().st_size, "bytes; shape", mats.shape,
          "sha256(materials)", hashlib.sha256(mats.tobytes()).hexdigest()[:16],
          "sha256(densities)", hashlib.sha256(dens.tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    main()
