"""Material registry: MC-GPU material number = 1-based rank after a stable sort by nominal density.

Ordering rule and the 22 identifiers restated from `cbctmc/mc/materials.py:112-119` and the
`cbctmc/assets/material_files/*__5_125kev.mcgpu` headers.
"""
from __future__ import annotations

import lzma
import os
import re
from pathlib import Path
from typing import Dict, List, Sequence

# (identifier, nominal density g/cm^3) sorted the way the reference sorts them:
# file names sorted alphabetically, then a stable sort by density.
_ALPHABETICAL = [
    ("acrylic", 1.18), ("adipose", 0.95), ("air", 0.0013), ("aluminium", 2.7), ("blood", 1.06),
    ("bone_020", 1.14), ("bone_050", 1.4), ("bone_100", 1.92), ("cartilage", 1.1), ("delrin", 1.42),
    ("glands_others", 1.03), ("h2o", 1.0), ("ldpe", 0.92), ("liver", 1.05), ("lung", 0.1),
    ("muscle_tissue", 1.05), ("pmp", 0.83), ("polystyrene", 1.03), ("red_marrow", 1.03),
    ("soft_tissue", 1.0), ("stomach_intestines", 1.04), ("teflon", 2.16),
]
MATERIALS_125KEV: Dict[str, float] = dict(sorted(_ALPHABETICAL, key=lambda kv: kv[1]))
MATERIAL_IDS: List[str] = list(MATERIALS_125KEV.keys())


def material_number(identifier: str) -> int:
    """1-based MC-GPU material number (cbctmc/mc/materials.py:19-36)."""
    return MATERIAL_IDS.index(identifier) + 1


def material_filename(identifier: str) -> str:
    return f"{identifier}__5_125kev.mcgpu"


def resolve_material_files(search_dirs: Sequence[os.PathLike], out_dir: os.PathLike,
                           needed: Sequence[str] | None = None) -> List[Path]:
    """Return the 22 material file paths in MC-GPU order, materialised under `out_dir`.

    Files are looked up in `search_dirs` as `<id>__5_125kev.mcgpu[.xz|.gz]`; `.xz` fixtures are
    decompressed into `out_dir`.  For a material that is not available AND not `needed`, a
    header-only stub (name + nominal density) is written: the engine -- like the reference,
    MC-GPU_v1.3.cu:2220-2233 -- reads only the density of materials no voxel uses.
    """
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    paths = []
    for ident in MATERIAL_IDS:
        name = material_filename(ident)
        found = None
        for d in search_dirs:
            d = Path(d)
            if (d / name).is_file():
                found = d / name
                break
            if (d / (name + ".gz")).is_file():
                found = d / (name + ".gz")
                break
            if (d / (name + ".xz")).is_file():
                target = out_dir / name
                if not target.is_file():
                    tmp = out_dir / (name + f".tmp{os.getpid()}")
                    with lzma.open(d / (name + ".xz"), "rb") as fi, open(tmp, "wb") as fo:
                        fo.write(fi.read())
                    os.replace(tmp, target)
                found = target
                break
        if found is None:
            if needed is not None and ident in needed:
                raise FileNotFoundError(f"material file for '{ident}' not found in {list(map(str, search_dirs))}")
            target = out_dir / ("stub_" + name)
            target.write_text(
                "#[MATERIAL DEFINITION FOR MC-GPU: header-only stub, material not used by any voxel]\n"
                f"#[MATERIAL NAME]\n# {ident}(stub)\n#[NOMINAL DENSITY (g/cm^3)]\n# {MATERIALS_125KEV[ident]}\n"
            )
            found = target
        paths.append(Path(found))
    return paths


def parse_nominal_density(filepath: os.PathLike) -> float:
    opener = lzma.open if str(filepath).endswith(".xz") else open
    with opener(filepath, "rt") as f:
        grab = False
        for line in f:
            if grab:
                return float(line.strip("# \n"))
            grab = "NOMINAL DENSITY" in line
    raise ValueError(f"no nominal density in {filepath}")
