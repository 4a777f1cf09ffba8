"""Voxelised geometries and the MC-GPU `.vox(.gz)` wire format.

Host-side mirror of the parts of `cbctmc/mc/geometry.py` the engine path needs:
`MCGeometry.save_mcgpu_geometry` / `create_mcgpu_geometry` (geo.py:462-477, :579-623: rot90(k=3) in
the x/y plane and swapped x/y spacing before writing, x fastest), `MCAirGeometry` (geo.py:626-639),
`MCCatPhan604Geometry` (recipe geo.py:902-1068), `pad_to_shape` (geo.py:340-374), `copy` / `warp`
(geo.py:375-439; the warp runs on the GPU through the engine), `MCCIRSPhantomGeometry` (geo.py:642-878:
bundled base geometry + tumour / line-pair inserts).  The CT -> material mapping pipeline of the reference
(`from_image`) needs its segmentation networks and is out of scope.  `MCThoraxLikeGeometry` is NOT a reference
class: it is the synthetic patient-like workload of SURVEY.md 8d (input 3) used by tests and benchmarks.
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

    def copy(self) -> "MCGeometry":
        g = self.__class__.__new__(self.__class__)
        MCGeometry.__init__(g, self.materials.copy(), self.densities.copy(), self.image_spacing)
        return g

    def warp(self, vector_field: np.ndarray, engine_context) -> "MCGeometry":
        """Nearest-neighbour warp by a dense displacement field [(1,) 3, x, y, z] in voxels, air outside the volume
        (`MCGeometry.warp`, geo.py:386-439: vroc's SpatialTransformer = identity grid + field ->
        `grid_sample(mode="nearest", align_corners=True)`).  Runs on the GPU of `engine_context` (an open
        `engine.Context`); there is no host implementation."""
        field = np.asarray(vector_field, dtype=np.float32)
        if field.ndim == 5:
            field = field[0]
        if field.ndim != 4 or field.shape[0] != 3 or field.shape[1:] != self.image_shape:
            raise ValueError(f"Expected vector_field of shape (3, {self.image_shape}), but got {field.shape=}")
        u = np.ascontiguousarray(np.transpose(field, (0, 3, 2, 1)))  # engine arrays are [z][y][x]; components stay (x, y, z)
        m, d = engine_context.warp_volume(np.transpose(self.materials, (2, 1, 0)), np.transpose(self.densities, (2, 1, 0)), u,
                                          material_number("air"), MATERIALS_125KEV["air"])
        return MCGeometry(np.transpose(m, (2, 1, 0)), np.transpose(d, (2, 1, 0)), self.image_spacing)

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


ASSETS = Path(__file__).resolve().parent / "assets"


class MCCIRSPhantomGeometry(MCGeometry):
    """CIRS dynamic thorax phantom (geo.py:642-878).  The base geometry is the reference's bundled segmented CT
    (305 x 300 x 152 voxels of 1 mm; air, lung-equivalent = h2o at 0.207, soft_tissue, red_marrow, bone_020/050/100),
    shipped here as `assets/geometries/base_cirs_geometry.npz` (same arrays, see oracle/gen_cirs_asset.py)."""

    @classmethod
    def from_base_geometry(cls) -> "MCCIRSPhantomGeometry":
        with np.load(ASSETS / "geometries" / "base_cirs_geometry.npz") as f:
            return cls(f["materials"], f["densities"], tuple(float(s) for s in f["image_spacing"]))

    @staticmethod
    def create_spherical_mask(radius: float, shape, sphere_center) -> np.ndarray:
        """Voxels with (x-cx)^2 + (y-cy)^2 + (z-cz)^2 <= r^2 on integer voxel indices (geo.py:749-761)."""
        x, y, z = (np.arange(n, dtype=np.float64) for n in shape)
        return ((x[:, None, None] - sphere_center[0]) ** 2 + (y[None, :, None] - sphere_center[1]) ** 2
                + (z[None, None, :] - sphere_center[2]) ** 2) <= radius ** 2

    @staticmethod
    def create_cirs_insert(shape, insert_center) -> np.ndarray:
        """The 30 mm tumour sphere with its 3 mm marker bore: a cylinder of radius 1.5 from the sphere centre to its
        upper pole (both ends inclusive) is cut out (geo.py:763-792)."""
        radius = 15.0
        c = np.asarray(insert_center, dtype=np.float64)
        mask = MCCIRSPhantomGeometry.create_spherical_mask(radius, shape, c)
        x, y, z = (np.arange(n, dtype=np.float64) for n in shape)
        zc = c[2] + radius / 2
        bore = (((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 <= 1.5 ** 2)
                & (z[None, None, :] >= zc - radius / 2) & (z[None, None, :] <= zc + radius / 2))
        mask[bore] = False
        return mask

    def place_insert(self, shift=(0, 0, 0), insert_center=(238, 141, 71)) -> "MCCIRSPhantomGeometry":
        """Soft-tissue tumour insert at `insert_center + shift` [voxels] (geo.py:865-878)."""
        mask = self.create_cirs_insert(self.image_shape, np.asarray(insert_center) + np.asarray(shift))
        g = self.copy()
        g.materials[mask] = material_number("soft_tissue")
        g.densities[mask] = np.float32(MATERIALS_125KEV["soft_tissue"])
        return g

    def place_line_pair_insert(self, gap: float = 4) -> "MCCIRSPhantomGeometry":
        """Resolution insert: the x axis is refined 4x (0.25 mm voxels), then 4 line pairs of aluminium / lung-equivalent
        slabs, each `gap` mm wide and 40 x 40 voxels across, are written ending at x = 238 mm (geo.py:794-863)."""
        g = self.copy()
        g.materials = np.repeat(g.materials, 4, axis=0)
        g.densities = np.repeat(g.densities, 4, axis=0)
        g.image_spacing = (0.25, 1.0, 1.0)
        gap_voxel = int(gap // g.image_spacing[0])
        pair = 2 * gap_voxel
        n_pairs, width = 4, 20
        cx = int(238 / g.image_spacing[0] - n_pairs / 2 * pair)
        cy, cz = int(141 / g.image_spacing[1]), int(71 / g.image_spacing[2])
        ys, zs = slice(cy - width, cy + width), slice(cz - width, cz + width)
        for i in range(n_pairs):
            a = cx + i * pair
            g.materials[a:a + gap_voxel, ys, zs] = material_number("aluminium")
            g.densities[a:a + gap_voxel, ys, zs] = np.float32(MATERIALS_125KEV["aluminium"])
            g.materials[a + gap_voxel:a + pair, ys, zs] = material_number("h2o")
            g.densities[a + gap_voxel:a + pair, ys, zs] = np.float32(0.207 * MATERIALS_125KEV["h2o"])
        return g

    def downsample(self, step: int) -> "MCCIRSPhantomGeometry":
        """Every `step`-th voxel at `step` times the spacing (reduced test phantoms; not a reference method)."""
        g = self.__class__.__new__(self.__class__)
        MCGeometry.__init__(g, self.materials[::step, ::step, ::step], self.densities[::step, ::step, ::step],
                            tuple(s * step for s in self.image_spacing))
        return g


class MCThoraxLikeGeometry(MCGeometry):
    """Synthetic patient-like thorax (input 3; BASELINE config 4 shape 512 x 512 x 256 at 1 mm): seeded
    ellipsoids carrying the tissue classes of the reference's material mapper at their NOMINAL densities (the reference
    maps every class to its nominal density, geo.py:72-74).  Body soft tissue with an adipose rim and a muscle layer,
    two lungs with vessels (blood), heart (blood + muscle wall), liver, stomach, spine (bone_100 shell, red_marrow core,
    cartilage discs), ribs (bone_050), sternum (bone_020), glands, and `n_nodules` seeded soft-tissue nodules in the
    lungs.  Lengths scale with the shape, so reduced shapes give the same anatomy at coarser voxels.

    `bone_texture=True` gives the bones the voxel-level texture the reference's `BoneMaterialMapper` produces on a real CT
    (geo.py:138-166): inside the bone segmentation every voxel is classed by its own HU value -- red_marrow below 150, bone_020
    up to 300, bone_050 above, bone_100 for the one-voxel outline above 300 -- here from a seeded, smoothed random HU field
    around each structure's nominal value, so neighbouring voxels of a rib or a vertebra differ in material.  The same switch applies
    the reference's `AirMaterialMapper` (geo.py:168-183: voxels below -900 HU become air) to a seeded HU field of the lungs around
    -820 HU: one lung voxel in eight is air, scattered through the parenchyma as on a real CT."""

    def __init__(self, shape=(512, 512, 256), image_spacing=(1.0, 1.0, 1.0), seed: int = 1234, n_nodules: int = 24, bone_texture: bool = False):
        rng = np.random.default_rng(seed)
        scale = [n / r for n, r in zip(shape, (512.0, 512.0, 256.0))]
        # coordinates in "1 mm voxels of the full-size phantom", origin at the volume centre
        ax = [(np.arange(n, dtype=np.float32) + np.float32(0.5) - np.float32(n / 2)) / np.float32(sc) for n, sc in zip(shape, scale)]
        mats = np.full(shape, material_number("air"), dtype=np.uint8)
        dens = np.full(shape, MATERIALS_125KEV["air"], dtype=np.float32)

        def ell(c, a, cut=None):
            """(slices, mask) of the ellipsoid centre c, half axes a, inside its bounding box."""
            sl = []
            for k in range(3):
                lo, hi = np.searchsorted(ax[k], c[k] - a[k], "left"), np.searchsorted(ax[k], c[k] + a[k], "right")
                sl.append(slice(int(lo), int(hi)))
            x, y, z = ax[0][sl[0]][:, None, None], ax[1][sl[1]][None, :, None], ax[2][sl[2]][None, None, :]
            m = ((x - c[0]) / a[0]) ** 2 + ((y - c[1]) / a[1]) ** 2 + ((z - c[2]) / a[2]) ** 2 <= 1.0
            if cut is not None:
                m = m & cut(x, y, z)
            return tuple(sl), m

        def put(c, a, ident, only=None, cut=None):
            """Fill the ellipsoid with `ident`; `only` restricts it to voxels currently holding that material."""
            sl, m = ell(c, a, cut)
            if only is not None:
                m = m & (mats[sl] == material_number(only))
            mats[sl][m] = material_number(ident)
            dens[sl][m] = np.float32(MATERIALS_125KEV[ident])

        BX, BY, INF = 225.0, 165.0, 1.0e4
        put((0, 0, 0), (BX, BY, INF), "adipose")
        put((0, 0, 0), (BX - 14, BY - 13, INF), "muscle_tissue")
        put((0, 0, 0), (BX - 27, BY - 25, INF), "soft_tissue")
        for k in range(-6, 7):  # ribs: thin elliptical shells in the soft tissue, interrupted at the front
            put((0, 0, k * 20.0), (BX - 29, BY - 27, 4.5), "bone_050", only="soft_tissue", cut=lambda x, y, z: y > -95)
            put((0, 0, k * 20.0), (BX - 40, BY - 37, 4.5), "soft_tissue", only="bone_050")
        for s in (-1, 1):
            put((s * 92, -10, 10), (74, 102, 115), "lung")
            for k in range(9):  # vessels
                c = (s * (92 + rng.uniform(-38, 38)), -10 + rng.uniform(-60, 60), 10 + rng.uniform(-80, 80))
                put(c, (rng.uniform(3, 6), rng.uniform(3, 6), rng.uniform(25, 60)), "blood", only="lung")
            for k in range(n_nodules // 2):
                c = (s * (92 + rng.uniform(-45, 45)), -10 + rng.uniform(-70, 70), 10 + rng.uniform(-90, 90))
                r = rng.uniform(4, 11)
                put(c, (r, r, r), "soft_tissue", only="lung")
        put((15, -36, 18), (54, 47, 48), "muscle_tissue")   # heart wall
        put((15, -36, 18), (43, 36, 39), "blood")           # chambers
        put((-70, -6, -95), (90, 80, 40), "liver", only="soft_tissue")
        put((78, -20, -98), (54, 47, 30), "stomach_intestines", only="soft_tissue")
        put((0, 105, 0), (26, 26, INF), "bone_100")         # vertebral column
        put((0, 105, 0), (17, 17, INF), "red_marrow")
        for k in range(-6, 7):
            put((0, 105, k * 20.0 + 10.0), (26, 26, 2.5), "cartilage")
        put((0, -128, 20), (26, 8, 70), "bone_020")         # sternum
        for s in (-1, 1):
            put((s * 122, -80, 60), (20, 13, 22), "glands_others", only="soft_tissue")
        if bone_texture:
            from scipy import ndimage
            nominal_hu = {"red_marrow": 90.0, "bone_020": 230.0, "bone_050": 420.0, "bone_100": 700.0}
            bone = np.zeros(shape, dtype=bool)
            hu = np.zeros(shape, dtype=np.float32)
            for ident, value in nominal_hu.items():
                m = mats == material_number(ident)
                bone |= m
                hu[m] = value
            sl = tuple(slice(int(lo), int(hi) + 1) for lo, hi in ((i.min(), i.max()) for i in np.nonzero(bone)))  # the bones' bounding box
            noise = ndimage.gaussian_filter(np.random.default_rng(seed + 1).standard_normal(hu[sl].shape).astype(np.float32), 0.8)
            hu_b = hu[sl] + np.float32(170.0 / float(noise.std())) * noise
            mask = bone[sl]
            outline = mask & ~ndimage.binary_erosion(mask)
            for ident, m in (("red_marrow", mask & (hu_b < 150)), ("bone_020", mask & (hu_b >= 150) & (hu_b < 300)),
                             ("bone_050", mask & (hu_b >= 300)), ("bone_100", outline & (hu_b >= 300))):
                mats[sl][m] = material_number(ident)
                dens[sl][m] = np.float32(MATERIALS_125KEV[ident])
            lung = mats == material_number("lung")
            sl = tuple(slice(int(i.min()), int(i.max()) + 1) for i in np.nonzero(lung))
            noise = ndimage.gaussian_filter(np.random.default_rng(seed + 2).standard_normal(mats[sl].shape).astype(np.float32), 1.0)
            air = lung[sl] & (np.float32(-820.0) + np.float32(70.0 / float(noise.std())) * noise < np.float32(-900.0))
            mats[sl][air] = material_number("air")
            dens[sl][air] = np.float32(MATERIALS_125KEV["air"])
        super().__init__(mats, dens, image_spacing)
