"""Workloads, their inputs, and the identity of the kernel build being measured."""
from __future__ import annotations

import hashlib
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def package():
    """The engine package (4d-cbct-mc_amd/, imported as cbctmc_amd)."""
    from __graft_entry__ import load_package
    return load_package()


def checker_paths():
    """tests/ and oracle/ hold the CHECKERS of the bench (the CPU oracle and the statistics that compare it with the kernel); only the
    legs that check or time the oracle put them on the import path -- `bench.py --no-cpu-baseline --no-compat` runs without them."""
    for d in ("tests",):
        if str(ROOT / d) not in sys.path:
            sys.path.insert(0, str(ROOT / d))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s

# Algorithmic bytes per history in the REFERENCE's table layout:
#   8 B x voxel gathers + 24 B x MFP rows + 8 B x Woodcock rows + 16 B x tally read-modify-writes,
# event counts per history measured by the instrumented oracle on each geometry (DESIGN.md 3.1 "Roofline").
WORKLOADS = {
    # name: (label, algorithmic bytes per history, where the figure comes from)
    "catphan": ("catphan604_{v}cube_1mm", 241.0, "SURVEY 8d: 8x21.26 + 24x1.84 + 8x1.47 + 16x0.93"),
    "cirs": ("cirs_305x300x152_1mm_insert", 149.0, "SURVEY 8d: 8x8.76 + 24x2.12 + 8x1.72 + 16x0.90"),
    "thorax": ("thorax_like_512x512x256_1mm", 356.0, "DESIGN 3.1: 8x24.12 + 24x5.36 + 8x2.93 + 16x0.69 (oracle counters, projection 0)"),
    # the thorax with the voxel-level bone texture of the reference's BoneMaterialMapper (geo.py:138-166)
    "thorax_textured": ("thorax_like_512x512x256_1mm_bone_and_lung_texture", 364.0, "8x23.98 + 24x5.76 + 8x2.90 + 16x0.68 (oracle counters, projection 0: profiles/r05w_*)"),
}
KERNEL_SOURCES = ("track_pool.inc", "track_common.inc", "device_model.hpp", "track_fast.hip")


def _code_only(text: str) -> bytes:
    """A source file without its comments and blank lines (none of the kernel sources holds a string literal with a comment marker):
    a corrected comment is not another kernel build."""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lines = [re.sub(r"//.*$", "", line).rstrip() for line in text.split("\n")]
    return "\n".join(line for line in lines if line.strip()).encode()


def kernel_source_hash() -> str:
    """Identifies the FAST kernel build: SHA-256 over the code of its sources (comments and blank lines removed) and over the compiler
    flags of track_fast.o (the CXXFLAGS / HIPFLAGS / FASTMATH lines of the Makefile)."""
    h = hashlib.sha256()
    csrc = ROOT / "4d-cbct-mc_amd" / "csrc"
    for name in KERNEL_SOURCES:
        h.update(_code_only((csrc / name).read_text()))
    for line in (csrc / "Makefile").read_text().split("\n"):
        if line.startswith(("CXXFLAGS", "HIPFLAGS", "FASTMATH")):
            h.update(line.encode())
    return h.hexdigest()[:16]

def build_workload(workdir: Path, workload, histories: int, n_proj: int, engine, n_vox: int = 512):
    """Geometry + input file in the reference's wire formats (written once, by rank 0): the package's own builder
    (4d-cbct-mc_amd/workloads.py: geometry.vox, the reference's text format, + the geometry.voxbin sidecar the engine prefers)."""
    return package().workloads.build_workload(workdir, workload, histories, n_proj, engine=engine, n_vox=n_vox)

def usable_cpus() -> int:
    """Host threads this process may actually use: scheduler affinity, capped by the cgroup CPU quota (a GPU box hands a
    1-GPU job a share of the host, while os.cpu_count() reports every core of the machine)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)

def knob_environment() -> dict:
    """The MCGPU_* tuning knobs of this process that select kernel variants and schedules at run time (scope K of the engine's
    registry, csrc/knobs.cpp -- the same list `MC-GPU_v1.3.x --knobs` prints); host-pipeline knobs, test hooks and the library
    path do not change the kernel and are not part of the stamp."""
    kernel_knobs = {k["name"] for k in package().engine.knob_table() if k["scope"] == "K"}
    return {k: v for k, v in sorted(os.environ.items()) if k in kernel_knobs}

def kernel_variant(workload: str, ctx=None) -> dict:
    """Which FAST kernel the engine dispatches for a workload: the template (tile records or plain u8 volume) and the scheduler are
    chosen when the model is uploaded (model_device.cpp), not by the kernel's sources."""
    if ctx is None:
        eng = package().engine
        wd = Path(tempfile.gettempdir()) / f"mcgpu_bench_{workload}_512_894"
        if not (wd / "input.in").exists():
            wd.mkdir(parents=True, exist_ok=True)
            build_workload(wd, workload, 100_000_000, 894, eng)
        with eng.create(str(wd / "input.in"), device=0) as c:
            return kernel_variant(workload, c)
    try:
        sched = int(ctx.geti("fast_scheduler"))
    except Exception:  # noqa: BLE001 -- a library of an earlier round (A/B runs through MCGPU_AMD_LIB) has the per-wave pools only
        sched = 0
    try:
        seg = int(ctx.geti("segment_loop"))
    except Exception:  # noqa: BLE001
        seg = 0
    return {"tile_records": int(ctx.geti("tile_records")), "fast_scheduler": sched, "segment_loop": seg, "volume_kind": int(ctx.geti("volume_kind"))}
