"""Voxelised geometries and the MC-GPU `.vox(.gz)` wire format.

Host-side mirror of the parts of `cbctmc/mc/geometry.py` the engine path needs:
`MCGeometry.save_mcgpu_geometry` / `create_mcgpu_geometry` (geo.py:462-477, :579-623: rot90(k=3) in
the x/y plane and swapped x/y spacing before writing, x fastest), `MCAirGeometry` (geo.py:626-639),
`MCCatPhan604Geometry` (recipe geo.py:902-1068), `pad_to_shape` (geo.py:340-374).  The CT -> material
mapping pipeline of the reference needs its segmentation networks and is out of scope.
"""
from __future__ import annotations

import gzip
from pathlib import Path
from typing import Dict, Sequence, Tuple

import numpy as np

from .materials import MATERIALS_125KEV, material_number


class MCGeometry:
    def __init__(self, materials: np.ndarray, densities: np.ndarray,
                 image_spacing: Tuple[float, float, float] = (1.0, 1.0, 1.0)):
        if materials.shape != densities.shape:
            raise ValueError(f"Shape mismatch: {materials.shape=} != {densities.shape=}")
        self.materials = np.ascontiguousarray(materials, dtype=np.uint8)
        self.densities = np.ascontiguousarray(densities, dtype=np.float32)
        self.image_spacing = tuple(float(s) for s in image_spacing)  # mm

    @property
    def image_shape(self) -> Tuple[int, int, int]:
        return self.materials.shape

    @property
    def image_size(self) -> Tuple[float, float, float]:
        return tuple(sh * sp for sh, sp in zip(self.image_shape, self.image_spacing))

    def pad_to_shape(self, target_shape: Sequence[int]) -> "MCGeometry":
        """Centre-pad with air (material 1, density 0.0013) up to `target_shape`."""
        air = MATERIALS_125KEV["air"]
        pads = []
        for have, want in zip(self.image_shape, target_shape):
            extra = max(int(want) - have, 0)
            pads.append((extra // 2, extra - extra // 2))
        mats = np.pad(self.materials, pads, mode="constant", constant_values=material_number("air"))
        dens = np.pad(self.densities, pads, mode="constant", constant_values=np.float32(air))
        return MCGeometry(mats, dens, self.image_spacing)

    def mcgpu_arrays(self):
        """(materials, densities, spacing_cm) exactly as written to the voxel file (x fastest)."""
        mats = np.rot90(self.materials, k=3, axes=(0, 1))
        dens = np.rot90(self.densities, k=3, axes=(0, 1))
        spacing_cm = (self.image_spacing[1] / 10.0, self.image_spacing[0] / 10.0, self.image_spacing[2] / 10.0)
        return mats, dens, spacing_cm

    def save_mcgpu_geometry(self, filepath, compress: bool = True, engine=None, binary_sidecar: bool = False):
        """Text voxel file of the reference (geo.py:579-623).  `binary_sidecar` (needs `engine`) also writes
        `<stem>.voxbin`, which the engine loads instead of parsing the text."""
        if not (self.densities > 0.0).all():
            raise ValueError("Density can not be zero or negative")
        mats, dens, spacing_cm = self.mcgpu_arrays()
        write_vox(filepath, mats, dens, spacing_cm, compress=compress, engine=engine)
        if binary_sidecar:
            if engine is None:
                raise ValueError("the binary sidecar is written by the engine library")
            nx, ny, nz = mats.shape
            engine.write_voxel_binary(engine.voxel_sidecar_path(filepath), (nx, ny, nz), spacing_cm,
                                      np.transpose(mats, (2, 1, 0)), np.transpose(dens, (2, 1, 0)))


def write_vox(filepath, materials_xyz: np.ndarray, densities_xyz: np.ndarray, spacing_cm, compress=True, engine=None):
    """Write a voxel file; arrays are indexed [x, y, z]; the file runs x fastest, then y, then z.

    With `engine` (the loaded C-ABI library wrapper) the multi-threaded C++ writer is used;
    otherwise a pure-Python writer (fine for the small test volumes).  Body format as
    `cbctmc/mc/voxel_data.pyx:12-31`: "<mat> <density:.6f>", blank line after each x-row and
    another after each slice.
    """
    filepath = str(filepath)
    nx, ny, nz = materials_xyz.shape
    m_lin = np.ascontiguousarray(np.transpose(materials_xyz, (2, 1, 0)), dtype=np.uint8)   # [z][y][x]
    d_lin = np.ascontiguousarray(np.transpose(densities_xyz, (2, 1, 0)), dtype=np.float32)
    if engine is not None:
        engine.write_voxel_file(filepath, (nx, ny, nz), spacing_cm, m_lin, d_lin, gzip=compress)
        return
    head = (
        "[SECTION VOXELS HEADER v.2008-04-13]\n"
        f"{nx} {ny} {nz}  # SIZE IN X, Y, Z\n"
        f"{spacing_cm[0]} {spacing_cm[1]} {spacing_cm[2]}  # VOXEL SPACING IN X, Y, Z\n"
        "1  # COLUMN NUMBER WHERE MATERIAL ID IS LOCATED\n2  # COLUMN NUMBER WHERE MASS DENSITY IS LOCATED\n"
        "1  # BLANK LINES AT END OF X,Y-CYCLES (1=YES, 0=NO)\n[END OF VXH SECTION]\n#\n"
    )
    parts = [head]
    for k in range(nz):
        for j in range(ny):
            row_m, row_d = m_lin[k, j], d_lin[k, j]
            parts.append("".join(f"{int(a)} {float(b):.6f}\n" for a, b in zip(row_m, row_d)))
            parts.append("\n")
        parts.append("\n")
    text = "".join(parts)
    if compress:
        with gzip.open(filepath, "wt", compresslevel=1) as f:
            f.write(text)
    else:
        with open(filepath, "wt") as f:
            f.write(text)


class MCAirGeometry(MCGeometry):
    """One 2000 mm voxel of air: the flat-field ("air") scan geometry (geo.py:626-639)."""

    def __init__(self, image_spacing=(2000.0, 2000.0, 2000.0)):
        super().__init__(np.full((1, 1, 1), material_number("air"), dtype=np.uint8),
                         np.full((1, 1, 1), MATERIALS_125KEV["air"], dtype=np.float32), image_spacing)


def _cylinder(shape, center, radius, height):
    """Boolean [x,y,z] mask: (x-cx)^2+(y-cy)^2 <= r^2 and cz-h/2 <= z < cz+h/2 on voxel indices."""
    x = np.arange(shape[0], dtype=np.float64)[:, None]
    y = np.arange(shape[1], dtype=np.float64)[None, :]
    disk = (x - center[0]) ** 2 + (y - center[1]) ** 2 <= radius ** 2
    z = np.arange(shape[2], dtype=np.float64)
    zsel = (z >= center[2] - height / 2) & (z < center[2] + height / 2)
    return disk, zsel


class MCCatPhan604Geometry(MCGeometry):
    """Catphan 604 sensitometry module in a water-filled body (recipe: geo.py:902-1068).

    angle [deg], distance/radius/length [voxels at 1 mm].  Later entries overwrite earlier ones.
    """

    BODY = [("h2o", 0.0, 0.0, 100.0, 100.0)]
    SENSITOMETRY = [
        ("air", 90, 58.7, 6.5, 24.0), ("teflon", 60, 58.7, 6.5, 24.0), ("delrin", 0, 58.7, 6.5, 24.0),
        ("bone_020", 330, 58.7, 6.5, 24.0), ("acrylic", 300, 58.7, 6.5, 24.0), ("air", 270, 58.7, 6.5, 24.0),
        ("polystyrene", 240, 58.7, 6.5, 24.0), ("ldpe", 180, 58.7, 6.5, 24.0), ("bone_050", 150, 58.7, 6.5, 24.0),
        ("pmp", 120, 58.7, 6.5, 24.0), ("h2o", 0, 0.0, 30.0, 40.0),
    ]
    CIRCULAR_SYMMETRY = [("air", 135, 35.355, 1.5, 24.0), ("air", 45, 35.355, 1.5, 24.0),
                         ("air", 315, 35.355, 1.5, 24.0), ("air", 225, 35.355, 1.5, 24.0)]

    def __init__(self, shape=(500, 500, 500), image_spacing=(1.0, 1.0, 1.0), scale: float = 1.0,
                 material_numbers: Dict[str, int] | None = None):
        """`scale` shrinks all lengths (voxel units) for reduced test phantoms; `material_numbers`
        overrides the MC-GPU material numbering (default: the reference's 22-material order)."""
        num = material_numbers or {k: material_number(k) for k in MATERIALS_125KEV}
        center = np.array(shape, dtype=np.float64) / 2
        mats = np.full(shape, num["air"], dtype=np.uint8)
        dens = np.full(shape, MATERIALS_125KEV["air"], dtype=np.float32)
        for group in (self.BODY, self.SENSITOMETRY, self.CIRCULAR_SYMMETRY):
            for ident, angle, distance, radius, length in group:
                phi = angle * np.pi / 180.0
                c = np.array([np.cos(phi), -np.sin(phi), 0.0]) * (distance * scale) + center
                disk, zsel = _cylinder(shape, c, radius * scale, length * scale)
                xs, ys = np.nonzero(disk)
                zs = np.nonzero(zsel)[0]
                if xs.size == 0 or zs.size == 0:
                    continue
                mats[xs[:, None], ys[:, None], zs[None, :]] = num[ident]
                dens[xs[:, None], ys[:, None], zs[None, :]] = np.float32(MATERIALS_125KEV[ident])
        super().__init__(mats, dens, image_spacing)


class MCBoxGeometry(MCGeometry):
    """Uniform block of one material (test geometry)."""

    def __init__(self, shape=(32, 32, 32), image_spacing=(5.0, 5.0, 5.0), material="h2o", density=None, number=None):
        n = number if number is not None else material_number(material)
        rho = MATERIALS_125KEV[material] if density is None else density
        super().__init__(np.full(shape, n, dtype=np.uint8), np.full(shape, rho, dtype=np.float32), image_spacing)
