"""The benchmark workloads of the engine (BASELINE.json configs 2-5) and the data assets they are built from.

What the reference ships under `cbctmc/assets/` and resolves in `cbctmc/mc/simulation.py:24-60` -- the 22 PENELOPE material
tables of 5-125 keV and the Varian 125 kVp spectrum -- lives under `assets/` of this package (the tables xz-compressed;
`resolve_material_files` unpacks them once into a cache directory).  `build_workload` writes a workload's geometry and input
file in the reference's wire formats, exactly what `MCSimulation.prepare_simulation` hands to `MC-GPU_v1.3.x`:

    catphan          Catphan604 in 512^3 voxels of 1 mm                         (config 2: the headline)
    cirs             the bundled CIRS thorax phantom with its tumour insert     (configs 3 and 5)
    thorax           synthetic 512 x 512 x 256 patient of 14 tissue classes     (config 4's shape)
    thorax_textured  the same with voxel-level bone texture and lung air       (what the reference's mappers make of a real CT)

`bench.py` and the measurement tools build their inputs here; `tests/` only checks.
"""
from __future__ import annotations

import os
import tempfile
from pathlib import Path
from typing import List

from . import geometry, materials, simulation

ASSETS = Path(__file__).resolve().parent / "assets"
WORKLOAD_NAMES = ("catphan", "cirs", "thorax", "thorax_textured")


def cache_dir() -> Path:
    """Where unpacked material tables go (MCGPU_TEST_CACHE, default /tmp/mcgpu_amd_test_cache: shared with the tests)."""
    return Path(os.environ.get("MCGPU_TEST_CACHE", os.path.join(tempfile.gettempdir(), "mcgpu_amd_test_cache")))


def material_files(raw_aluminium: bool = False) -> List[Path]:
    """The 22 material files in MC-GPU order.  The reference's aluminium table writes the integer columns ITL/ITU (and
    KZCO/KSCO) as "1.0 4.0", on which the reference's unchecked `sscanf("%d %d")` leaves ITU uninitialised
    (MC-GPU_v1.3.cu:2387-2392; docs/history.md section 2, deviation 10).  So that the reference build and the engine read the
    SAME numbers, the default is a copy with those columns rewritten as integers; `raw_aluminium=True` gives the file as
    shipped (tests/test_formats_and_abi.py checks that both parse to the same tables)."""
    paths = materials.resolve_material_files([ASSETS / "materials"], cache_dir() / "materials")
    if raw_aluminium:
        return paths
    k = materials.material_number("aluminium") - 1
    fixed = paths[k].with_name("aluminium__5_125kev.intcols.mcgpu")
    if not fixed.is_file():
        out, section = [], None
        for line in paths[k].read_text().split("\n"):
            if line.startswith("#"):
                section = "rita" if "COMMON/CGRA/" in line else "shells" if "COMMON/CGCO/" in line else section
            elif line.strip() and section in ("rita", "shells"):
                t = line.split()
                keep = 4 if section == "rita" else 3
                line = " ".join(t[:keep] + [str(int(float(v))) for v in t[keep:]])
            out.append(line)
        tmp = fixed.with_name(fixed.name + f".tmp{os.getpid()}")
        tmp.write_text("\n".join(out))
        os.replace(tmp, fixed)
    paths[k] = fixed
    return paths


def spectrum_file() -> Path:
    """The reference's default source spectrum (cbctmc/defaults.py:95)."""
    return ASSETS / "spectra" / "125kVp_0.89mmTi_varian_norm.spc"


def workload_geometry(workload: str, n_vox: int = 512):
    if workload == "catphan":
        return geometry.MCCatPhan604Geometry(shape=(n_vox,) * 3, image_spacing=(1.0, 1.0, 1.0))
    if workload == "cirs":
        return geometry.MCCIRSPhantomGeometry.from_base_geometry().place_insert()
    if workload == "thorax":
        return geometry.MCThoraxLikeGeometry()
    if workload == "thorax_textured":
        return geometry.MCThoraxLikeGeometry(bone_texture=True)
    raise ValueError(f"unknown workload {workload!r}: one of {WORKLOAD_NAMES}")


def workload_dir(workload: str, n_vox: int = 512, n_proj: int = 894) -> Path:
    """Default scratch directory of a workload's inputs (bench.py and the tools share it)."""
    return Path(tempfile.gettempdir()) / f"mcgpu_bench_{workload}_{n_vox}_{n_proj}"


def build_workload(workdir, workload: str, histories: int, n_proj: int, engine=None, n_vox: int = 512, binary_sidecar: bool = True) -> Path:
    """Geometry + input file of `workload` under `workdir` in the reference's wire formats: `geometry.vox` (text, what
    `MCGeometry.save_mcgpu_geometry` of the reference writes) plus, with `binary_sidecar`, `geometry.voxbin` (the engine prefers
    it: no 134 M-line parse).  Returns the path of `input.in`."""
    sim = simulation.MCSimulation(workload_geometry(workload, n_vox), material_files(), spectrum_file(), n_histories=histories,
                                  n_projections=n_proj, angle_between_projections=360.0 / n_proj)
    return sim.prepare_simulation(Path(workdir), compress_geometry=False, engine=engine, binary_sidecar=binary_sidecar)
