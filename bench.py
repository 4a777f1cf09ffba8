#!/usr/bin/env python3
"""bench.py -- photon histories/s of the MC CBCT projection hot path on MI355X.

Workload (BASELINE.json configs[1], `--workload catphan`, the default): Catphan604 phantom in a 512^3 volume @ 1 mm, Varian
half-fan geometry, 1848x768 detector, 894-projection trajectory, 1e8 histories per projection per GPU, default spectrum, real
PENELOPE material tables.  `--workload cirs` / `thorax` run the same measurement on the bundled CIRS phantom (configs 3/5
geometry) and on the patient-like 512x512x256 thorax (config 4 shape); they are not the headline.

A "step" is one projection: the photon-history kernel over one batch of histories (FAST personality), plus -- for N > 1 -- the
sum of the per-rank detector tallies (the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019) through the engine's tally exchange
(4d-cbct-mc_amd/csrc/exchange.cpp: copy-engine pushes into the owner's landing buffer beside the next projection's kernel,
one fused add on the owner; `BENCH_EXCHANGE=rccl` runs the plain RCCL reduction instead).  Inputs (volume, tables) are
resident in HBM before the timed region.  Weak scaling: every rank simulates `--histories` histories of each projection
with its own disjoint history-id range.

Ranks: one process per GPU.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) the
ranks exist already (RANK / LOCAL_RANK / WORLD_SIZE in the environment).  Without one, `python bench.py --gpus N` starts the
N rank processes ITSELF (before this process has touched the GPU or imported anything that could), waits for them and relays
rank 0's line -- like the reference's `mpirun -n N` inside `MCSimulation.run_simulation` (cbctmc/mc/simulation.py:187-198,
MC-GPU_v1.3.cu:389-391).  A rank count that cannot be honoured (fewer GPUs than ranks) exits non-zero; `BENCH_SHARE_GPU=1`
lets the ranks share the GPUs there are (development: two ranks on a one-GPU box exercise the whole N > 1 path).

Prints ONE JSON line on rank 0 (driver contract) carrying `roofline` and `cpu_baseline`.  At N = 1 the line also carries
driver-timed legs measured after the timed region: `compat` (the bit-exact personality), `check` (FAST against the oracle;
a failed check exits non-zero), `end_to_end` (the whole 894-projection scan with the MetaImage stacks on disk),
`end_to_end_ascii` (the drop-in default: the reference's 63 MB text file per projection) and `workloads` (the same kernel
measurement on the CIRS phantom and the patient-like thorax, configs 3-5).  At N > 1: `reduce` (bytes, copy-engine time,
the exposed add) and `check.sharded_equals_single` (the summed sharded tally against one rank simulating the same history
ids alone, bit for bit).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s

# Algorithmic bytes per history in the REFERENCE's table layout:
#   8 B x voxel gathers + 24 B x MFP rows + 8 B x Woodcock rows + 16 B x tally read-modify-writes,
# event counts per history measured by the instrumented oracle on each geometry (DESIGN.md 3.1 "Roofline").
WORKLOADS = {
    # name: (label, algorithmic bytes per history, where the figure comes from)
    "catphan": ("catphan604_{v}cube_1mm", 241.0, "SURVEY 8d: 8x21.26 + 24x1.84 + 8x1.47 + 16x0.93"),
    "cirs": ("cirs_305x300x152_1mm_insert", 149.0, "SURVEY 8d: 8x8.76 + 24x2.12 + 8x1.72 + 16x0.90"),
    "thorax": ("thorax_like_512x512x256_1mm", 356.0, "DESIGN 3.1: 8x24.12 + 24x5.36 + 8x2.93 + 16x0.69 (oracle counters, projection 0)"),
}
KERNEL_SOURCES = ("track_pool.inc", "track_common.inc", "device_model.hpp", "track_fast.hip")


def kernel_source_hash() -> str:
    """Identifies the FAST kernel build: SHA-256 over its sources and over the compiler flags of track_fast.o (the
    CXXFLAGS / HIPFLAGS / FASTMATH lines of the Makefile)."""
    h = hashlib.sha256()
    csrc = ROOT / "4d-cbct-mc_amd" / "csrc"
    for name in KERNEL_SOURCES:
        h.update((csrc / name).read_bytes())
    for line in (csrc / "Makefile").read_text().split("\n"):
        if line.startswith(("CXXFLAGS", "HIPFLAGS", "FASTMATH")):
            h.update(line.encode())
    return h.hexdigest()[:16]


def build_workload(workdir: Path, workload, histories: int, n_proj: int, engine, n_vox: int = 512):
    """Geometry + input file in the reference's wire formats (written once, by rank 0)."""
    import cases
    pkg = cases.pkg
    if workload == "catphan":
        geo = pkg.geometry.MCCatPhan604Geometry(shape=(n_vox,) * 3, image_spacing=(1.0, 1.0, 1.0))
    elif workload == "cirs":
        geo = pkg.geometry.MCCIRSPhantomGeometry.from_base_geometry().place_insert()
    elif workload == "thorax":
        geo = pkg.geometry.MCThoraxLikeGeometry()
    else:
        raise SystemExit(f"unknown workload {workload}")
    sim = pkg.simulation.MCSimulation(geo, cases.material_files(), cases.spectrum_file(), n_histories=histories, n_projections=n_proj,
                                      angle_between_projections=360.0 / n_proj)
    # geometry.vox (the reference's text format) + geometry.voxbin (binary sidecar the engine prefers: no 134 M-line parse)
    return sim.prepare_simulation(workdir, compress_geometry=False, engine=engine, binary_sidecar=True)


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


def cpu_baseline(ctx, seconds_budget: float = 16.0):
    """The restated CPU oracle (oracle/mcgpu_oracle.c, LIBM math == reference arithmetic; the loop being timed is the
    reference's MC-GPU_v1.3.cu:913-958) on a bounded sample of the same workload, on this host's cores.  Reported, never the
    target.  Also returns the sample's image and its sum of squared weights (for `check`)."""
    import oracle_lib as ol
    import parity
    T = parity.tables_from_context(ctx)
    usable = usable_cpus()
    hpt = 150
    image = np.zeros(T.image_size(), dtype=np.uint64)
    w2 = np.zeros(T.image_size(), dtype=np.uint64)
    cnt = ol.OracleCounters()
    batch0 = [0]

    def timed(threads, target_s):
        """Rate with `threads` OpenMP threads: chunks sized from the previous one until `target_s` of timed work."""
        nb, done, elapsed = max(threads * 16, 64), 0, 0.0
        while elapsed < target_s:
            t0 = time.perf_counter()
            T.track(0, 42, batch0[0], nb, hpt, ol.MATH_LIBM, n_threads=threads, image=image, counters=cnt, w2=w2)
            dt = time.perf_counter() - t0
            batch0[0] += nb
            if dt > 0.3 or threads == 1:  # chunks too short to time OpenMP start-up fairly are warm-up only
                done += nb
                elapsed += dt
            nb = int(max(threads * 16, min(nb * 1.5 / max(dt, 1e-3), 4e6)))
        return done * hpt / elapsed, done * hpt, elapsed

    curve = {}
    points = sorted({t for t in (1, 4, 16, 64, usable) if t <= usable})
    share = seconds_budget / (len(points) + 1)
    for t in points:
        rate, n, secs = timed(t, share * (2.0 if t == usable else 1.0))
        curve[str(t)] = rate
    c = cnt.as_dict()
    h = float(c["histories"])
    per_hist = {k: round(c[k] / h, 4) for k in ("steps", "voxel_reads", "mfp_reads", "woodcock_reads", "compton", "rayleigh", "photo", "rng", "tally_calls", "tally_hits")}
    algo = 8 * per_hist["voxel_reads"] + 24 * per_hist["mfp_reads"] + 8 * per_hist["woodcock_reads"] + 16 * per_hist["tally_calls"]
    out = {
        "value": curve[str(usable)], "unit": "histories/s", "cores": usable, "kind": "port",
        "sample": f"{n} histories of projection 0 of the same workload in {secs:.1f} s on {usable} threads (of {int(h)} in the whole thread curve), "
                  "OpenMP over RANECU batches (oracle/mcgpu_oracle.c, libm math)",
        "per_core_value": curve["1"], "threads_curve_histories_per_s": curve,
        "host": {"os_cpu_count": os.cpu_count(), "sched_affinity": len(os.sched_getaffinity(0)), "usable": usable},
        "events_per_history": per_hist, "algorithmic_bytes_per_history_from_these_counts": round(algo, 1),
    }
    return out, image, w2.astype(np.float64) * (1024.0 ** 2), int(h)


def pmc_summary(workload: str):
    """The committed rocprofv3 PMC summary of the FAST kernel (separate --pmc passes, tools/pmc_collect.sh), accepted only if
    it was collected from THIS kernel build (source hash) on THIS workload; else (None, reason)."""
    f = ROOT / "profiles" / ("pmc_summary_latest.json" if workload == "catphan" else f"pmc_summary_{workload}.json")
    if not f.exists():
        return None, "no summary committed"
    d = json.loads(f.read_text())
    stamp = d.get("_stamp", {})
    if stamp.get("kernel_source_sha16") != kernel_source_hash():
        return None, f"stale: summary is of kernel build {stamp.get('kernel_source_sha16')}, running {kernel_source_hash()}"
    if stamp.get("workload") != workload:
        return None, f"stale: summary is of workload {stamp.get('workload')}"
    return d, f"{f.name} ({stamp.get('collected', '?')})"


def compat_leg(ctx, torch, H, launches=3):
    """COMPAT personality (RANECU leap-frog streams, portable restatement of the reference's arithmetic, bit-identical to the
    oracle's portable mode) timed like the
    FAST steps: same projection schedule, the reference's launch shape for H histories (MC-GPU_v1.3.cu:824-841)."""
    nz, nx = ctx.detector_shape
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    batches, hpt, total = ctx.reference_shape(H)
    seed = ctx.geti("seed")
    ctx.launch(0, image.data_ptr(), batches, mode="compat", seed=seed, hpt=hpt, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch(((i + 1) * 149) % ctx.num_projections, image.data_ptr(), batches, mode="compat", seed=seed, hpt=hpt, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": total * launches / dt, "unit": "histories/s", "launches": launches, "histories_per_launch": total, "ms_per_launch": dt / launches * 1e3,
            "what": "COMPAT kernel: RANECU streams + a portable restatement of the reference arithmetic (own log/pow/sincos, glibc's expf): tallies bit-identical to the CPU oracle's portable mode (tests/test_gpu_fullsize.py), which differs from the reference build on <= 0.2 % of the tally words (last-bit differences of logf; tests/test_gpu_parity.py::test_compat_kernel_against_the_reference_build_itself)"}


def fast_vs_compat_check(ctx, runs=12, histories=250_000_000, projection=447):
    """FAST against the bit-exact COMPAT personality (tallies bit-identical to the oracle, tests/test_gpu_parity.py): `runs`
    independent launches of `histories` per mode, variances from the run-to-run scatter.  Detected energy per history per
    scatter class: ratio, relative sigma, z.  (tools/fast_vs_compat.py is the long version)"""
    p = projection % ctx.num_projections
    batches, hpt, _ = ctx.reference_shape(histories)
    ef, ec = [], []
    edge = {"fast": [0.0, 0.0, 0.0], "compat": [0.0, 0.0, 0.0]}  # primary energy: all columns, column 1024, beyond column 1024
    half_fan = ctx.detector_shape[1] == 1848  # the beam ends at the right edge of column 1023 (the reference crops there, projection.py:42-51)

    def note_edge(key, img):
        if half_fan:
            edge[key][0] += float(img[0].sum(dtype=np.float64)); edge[key][1] += float(img[0][:, 1024].sum(dtype=np.float64))
            edge[key][2] += float(img[0][:, 1025:].sum(dtype=np.float64))
    for k in range(runs):
        img, _, d = ctx.run_projection(p, histories, mode="fast", seed=8000 + k)
        ef.append(img.reshape(4, -1).sum(axis=1, dtype=np.float64) / d)
        note_edge("fast", img)
        img, _, d = ctx.run_projection(p, batches, mode="compat", seed=9000 + 7 * k, hpt=hpt)
        ec.append(img.reshape(4, -1).sum(axis=1, dtype=np.float64) / d)
        note_edge("compat", img)
    ef, ec = np.array(ef), np.array(ec)
    known = None
    if half_fan and edge["fast"][0] > 0 and edge["compat"][0] > 0:
        # KNOWN DEVIATION 1: the primary beam ends exactly at detector column 1024; photons within a hundredth of a
        # pixel of that edge fall to either side depending on the last bits of the sampled direction, and FAST (v_sin / v_cos) puts
        # more of them into column 1024 than the reference arithmetic.  Bounded: excess <= 5e-8 of the primary energy, nothing beyond.
        ff, cf = edge["fast"][1] / edge["fast"][0], edge["compat"][1] / edge["compat"][0]
        known = {"what": "primary energy in detector column 1024 (first column beyond the half-fan beam edge), fraction of the primary energy",
                 "fast": ff, "compat": cf, "excess": ff - cf, "bound_on_excess": 5e-8, "fast_beyond_column_1024": edge["fast"][2],
                 "compat_beyond_column_1024": edge["compat"][2],
                 "passed": bool(ff - cf <= 5e-8 and edge["fast"][2] == 0.0)}
    se = np.sqrt(ef.var(axis=0, ddof=1) / runs + ec.var(axis=0, ddof=1) / runs)
    z = (ef.mean(axis=0) - ec.mean(axis=0)) / np.where(se > 0, se, 1.0)
    return {"projection": int(p), "runs_per_mode": runs, "histories_per_run": int(histories), "classes": ["primary", "compton", "rayleigh", "multiple"],
            "energy_ratio_fast_over_compat": [float(a / b) if b else None for a, b in zip(ef.mean(axis=0), ec.mean(axis=0))],
            "relative_sigma": [float(a / b) if b else None for a, b in zip(se, ec.mean(axis=0))],
            "energy_z": [round(float(v), 3) for v in z], "beam_edge_column": known,
            "passed": bool(np.all(np.abs(z) < 6.0) and (known is None or known["passed"]))}  # Student t with 2 runs - 2 = 22 degrees of freedom: P(|t| > 6) = 5e-6 per class


def entry_face_deficit(ctx, runs=8, histories=1_000_000_000, projection=600):
    """KNOWN DEVIATION 2: the reference puts an entering photon EPS_SOURCE = 1.5e-5 cm past the entry face ALONG ITS
    RAY and calls everything within EPS_SOURCE of a face "outside" (MC-GPU_kernel_v1.3.cu:714-805, 1036-1042), so at oblique
    projections a first Woodcock step shorter than ~1.6e-5 cm is tallied at once as an un-attenuated primary: 5-7e-6 of the
    incident energy.  FAST's source_entry lands photons inside the object box and has no such shell (MCGPU_EXTERIOR_MODE=1 takes
    the reference's route).  Measured here on this context: primary of mode 1 over the default, `runs` launches each; the
    deficit must lie in [0, 1e-4] of the primary (within 4 sigma of the run-to-run scatter)."""
    p = projection % ctx.num_projections
    res = {}
    prev = os.environ.get("MCGPU_EXTERIOR_MODE")
    try:
        for mode_name, env in (("default", None), ("reference_entry", "1")):
            if env is None:
                os.environ.pop("MCGPU_EXTERIOR_MODE", None)
            else:
                os.environ["MCGPU_EXTERIOR_MODE"] = env
            ctx.reload_env_knobs()
            e = []
            for k in range(runs):
                img, _, d = ctx.run_projection(p, histories, mode="fast", seed=12000 + k)
                e.append(float(img[0].sum(dtype=np.float64)) / d)
            res[mode_name] = np.array(e)
    finally:
        if prev is None:
            os.environ.pop("MCGPU_EXTERIOR_MODE", None)
        else:
            os.environ["MCGPU_EXTERIOR_MODE"] = prev
        ctx.reload_env_knobs()
    a, b = res["default"], res["reference_entry"]
    deficit = float(1.0 - a.mean() / b.mean())
    sigma = float(np.sqrt(a.var(ddof=1) / runs + b.var(ddof=1) / runs) / b.mean())
    return {"what": "primary energy per history, 1 - default / MCGPU_EXTERIOR_MODE=1 (the reference's entry-face shell)", "projection": int(p),
            "runs_per_mode": runs, "histories_per_run": int(histories), "deficit": deficit, "sigma": sigma, "bounds": [0.0, 1e-4],
            "passed": bool(-4.0 * sigma <= deficit <= 1e-4 + 4.0 * sigma)}


def oracle_check(ctx, H, img_cpu, w2_cpu, n_cpu):
    """FAST vs the oracle sample of cpu_baseline on projection 0: detected energy per history per scatter class (ratio and
    z with the oracle's measured variance) and 16x16-pixel blocks."""
    import parity
    img_gpu, _, done = ctx.run_projection(0, H, mode="fast", seed=4242)
    img_cpu, w2_cpu = img_cpu.reshape(img_gpu.shape), w2_cpu.reshape(img_gpu.shape)
    zs = parity.class_energy_z(img_gpu, done, img_cpu, w2_cpu, n_cpu)
    ratio = [float(img_gpu[k].sum() / done / (img_cpu[k].sum() / n_cpu)) if img_cpu[k].sum() else None for k in range(4)]
    z, mask = parity.measured_z(parity.blocks(img_gpu, 16), done, parity.blocks(img_cpu, 16), parity.blocks(w2_cpu, 16), n_cpu)
    zz = z[mask]
    return {"projection": 0, "fast_histories": int(done), "oracle_histories": int(n_cpu), "classes": ["primary", "compton", "rayleigh", "multiple"],
            "energy_ratio_fast_over_oracle": ratio, "energy_z": [None if not np.isfinite(v) else round(v, 3) for v in zs],
            "blocks_16x16": int(mask.sum()), "blocks_beyond_3_sigma": float(np.mean(np.abs(zz) > 3.0)) if zz.size else None,
            "blocks_z_mean": float(zz.mean()) if zz.size else None, "blocks_z_std": float(zz.std()) if zz.size else None,
            "passed": bool(all((not np.isfinite(v)) or abs(v) < 4.0 for v in zs) and (zz.size == 0 or np.mean(np.abs(zz) > 3.0) < 0.01))}


def end_to_end_scan(ctx, H, workdir, n=894, ascii_files=False):
    """The pipelined scan driver (track -> finalize -> pinned copy -> writer thread) over `n` projections with its output on
    disk: the three MetaImage stacks, or (`ascii_files`) the reference's ASCII file per projection -- 63 MB of text each,
    formatted on the device (the unchanged-cbctmc drop-in default).  Per-projection wall time including output."""
    out = workdir / ("scan_ascii" if ascii_files else "scan_out")
    out.mkdir(exist_ok=True)
    crop = 1024 if ctx.detector_shape[1] == 1848 else 0
    first = min(100, max(ctx.num_projections - n, 0))
    rep = ctx.run_scan(mode="fast", first_projection=first, num_projections=n, histories=H, crop_nx=crop, write_stacks=not ascii_files,
                       write_ascii=ascii_files, output_folder=out, pixel_spacing=(0.776, 0.776))
    for f in out.glob("projections_*.mha"):
        f.unlink()
    res = {"projections": n, "seconds_total": rep["seconds_total"], "ms_per_projection_kernels": rep["seconds_kernels"] / n * 1e3,
           "writer_ms_per_projection": rep["seconds_writer"] / n * 1e3, "drain_after_last_kernel_ms": rep["seconds_after_last_kernel"] * 1e3}
    if ascii_files:
        files = [Path(ctx.projection_file_name(p)) for p in range(first, first + n)]
        res["file_bytes_mean"] = float(np.mean([f.stat().st_size for f in files if f.exists()]))
        res["files_written"] = int(sum(f.exists() for f in files))
        for f in files:
            f.unlink(missing_ok=True)
        res["histories_per_s_with_ascii_files"] = n * H / rep["seconds_total"]
        res["ms_per_projection_with_ascii_files"] = rep["seconds_total"] / n * 1e3
    else:
        res["histories_per_s_with_stacks"] = n * H / rep["seconds_total"]
        res["ms_per_projection_with_stacks"] = rep["seconds_total"] / n * 1e3
    return res


def timed_launches(ctx, torch, H, launches=8, warm=2):
    """Mean kernel time [ms] of `launches` FAST launches of H histories (HIP events on the launch stream), after `warm` untimed."""
    nz, nx = ctx.detector_shape
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed, nproj, ms = ctx.geti("seed"), ctx.num_projections, []
    for i in range(warm + launches):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch((i * 149) % nproj, image.data_ptr(), H, mode="fast", seed=seed, stream=stream)
        t = ctx.last_kernel_ms()
        if i >= warm:
            ms.append(t)
    return float(np.mean(ms)), float(np.min(ms)), int(image.sum().item())


def measured_ceilings(ctx):
    """The two hardware ceilings the FAST kernel is priced against, measured NOW on this GPU by the library's micro-benchmarks
    (mcgpu_microbench, csrc/microbench.hip; about 20 ms each): vector-instruction issue of a dense dependent-FMA kernel at 8
    waves/SIMD under three EXEC masks, and scattered 64-bit atomic adds into a detector-sized tally."""
    v = ctx.microbench("valu_issue")
    return {"valu_wave_instructions_per_ns_per_simd": {"64_active_lanes": v[0], "lanes_0_31": v[1], "32_lanes_spread": v[2]},
            "scattered_64bit_atomic_adds_per_s": ctx.microbench("atomic_rate"), "source": "mcgpu_microbench in this run"}


def roofline_block(workload, H, k_ms, ceilings=None, ctx=None):
    """`roofline` object of one workload: algorithmic bytes of the reference layout over the measured kernel time, plus the
    PMC-counter traffic of this kernel build when a stamped summary of it is committed."""
    label, algo_bytes, algo_src = WORKLOADS[workload]
    achieved = algo_bytes * H / (k_ms * 1e-3) / 1e9
    pmc, pmc_src = pmc_summary(workload) if H == int(1e8) else (None, "summary is per 1e8-history launch")
    traffic = hbm_counter_frac = valu = l2_hit = None
    if pmc:
        # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled
        # (MI355X_MICROARCH.md, HBM section)
        traffic = (2.0 * pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
        hbm_counter_frac = traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        insts = pmc["SQ_INSTS_VALU"]["mean_per_dispatch"]
        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
            l2_hit = pmc["TCC_HIT_sum"]["mean_per_dispatch"] / max(pmc["TCC_HIT_sum"]["mean_per_dispatch"] + pmc["TCC_MISS_sum"]["mean_per_dispatch"], 1.0)
        # ceiling: a dense dependent-FMA kernel, 8 waves/SIMD, 16-32 active lanes, on the same chip (tools/archive/micro/exec_skip.hip:
        # 5.24e9 wave-instructions on 1024 SIMDs in 5.1 ms)
        # measured in this run when `ceilings` is given (the higher of the half-populated masks: the conservative peak); else the
        # builder-run figure of round 3 (5.24e9 wave-instructions on 1024 SIMDs in 5.1 ms)
        if ceilings:
            cv = ceilings["valu_wave_instructions_per_ns_per_simd"]
            peak, peak_src = max(cv["lanes_0_31"], cv["32_lanes_spread"], cv["64_active_lanes"]), "measured in this run (mcgpu_microbench)"
        else:
            peak, peak_src = 5.24e9 / 1024.0 / 5.1e6, "tools/archive/micro/exec_skip.hip, round 3"
        lane_util = pmc["SQ_THREAD_CYCLES_VALU"]["mean_per_dispatch"] / pmc["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / 64.0
        valu = {"valu_wave_instructions_per_launch": insts, "valu_wave_instructions_per_history": insts / H,
                "achieved_per_ns_per_simd": insts / 1024.0 / (k_ms * 1e6),
                "measured_peak_per_ns_per_simd": peak, "peak_source": peak_src, "frac": insts / 1024.0 / (k_ms * 1e6) / peak,
                "lane_utilisation": lane_util}
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "hbm_counter_frac": hbm_counter_frac, "traffic_source": pmc_src,
            "fabric_bytes_per_history": None if traffic is None else traffic / H, "l2_hit_rate": l2_hit,
            "kernel": "track_pool_kernel<4> (fast, u8 volume + tile records)" if (ctx is not None and ctx.geti("tile_records")) else "track_pool_kernel<0> (fast, u8 volume)",
            "kernel_ms_avg": k_ms, "kernel_source_sha16": kernel_source_hash(),
            "algorithmic_bytes_per_history": algo_bytes, "algorithmic_bytes_source": algo_src,
            "algorithmic_bytes_per_launch": algo_bytes * H}
    if valu:
        # what binds the launch: `frac` above is the contract's model figure (reference-layout bytes over the kernel
        # time), NOT the HBM utilisation (that is hbm_counter_frac); the resource that is actually scarce is vector lane-slots
        roof["binding"] = {"resource": "valu lane-slots", "frac": valu["frac"] * valu["lane_utilisation"],
                           "issue_frac": valu["frac"], "lane_utilisation": valu["lane_utilisation"],
                           "note": "vector-instruction issue rate over the measured dense-FMA ceiling, times the fraction of lanes active in an issued instruction"}
    return roof, valu


def cirs_4d_leg(c2, torch, H, states=10, projections_per_state=89):
    """Config 5 (cbctmc/mc/simulation.py:527-710): `states` respiratory states of the CIRS phantom, each a 167 MB displacement
    field uploaded and applied ON THE DEVICE (mcgpu_warp_geometry: warp of the index volume, brick grids, object box, majorant),
    followed by `projections_per_state` projections of H histories in the warped geometry."""
    nz, nx = c2.detector_shape
    shape = (c2.geti("num_voxels_y"), c2.geti("num_voxels_x"), c2.geti("num_voxels_z"))  # frame of the MCGeometry arrays (engine.warp_geometry)
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed = c2.geti("seed")
    zz = np.linspace(-1, 1, shape[2], dtype=np.float32)[None, None, :]
    field = np.zeros((3,) + shape, np.float32)
    warp_s = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in range(states):
        field[2] = (15.0 * np.sin(2 * np.pi * st / states)) * (1 - zz * zz)  # SI motion up to 15 mm (SURVEY 8d input 4)
        torch.cuda.synchronize()  # the previous state's projections are done before the geometry changes under them
        tw = time.perf_counter()
        c2.warp_geometry(field, frame="geometry")
        warp_s.append(time.perf_counter() - tw)
        for k in range(projections_per_state):
            c2.clear(image.data_ptr(), stream)
            c2.launch((st * projections_per_state + k) % c2.num_projections, image.data_ptr(), H, mode="fast", seed=seed, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c2.warp_geometry(np.zeros((3,) + shape, np.float32), frame="geometry")  # back to the base geometry
    return {"states": states, "projections_per_state": projections_per_state, "field_bytes": int(field.nbytes), "seconds_total": dt,
            "ms_per_state_change": float(np.mean(warp_s[1:]) * 1e3), "ms_first_state_change": warp_s[0] * 1e3,
            "value": states * projections_per_state * H / dt, "unit": "histories/s",
            "what": "device-side respiratory states (field upload + warp + brick grids + majorant) followed by their projections, one stream"}


def fdk_leg(pkg, device, n=894, nu=1024, nv=768, du=0.388, pad=1.0):
    """Config 4's reconstruction (cbctmc/reconstruction/reconstruction.py:22-69: rtkfdk --pad 1 --hann 1 --hannY 1, 464 x 250 x
    464 voxels of 1 mm) through the in-process FDK (csrc/fdk.hip, parity unpinned against RTK): synthetic projections of the
    reference's size, kernel times from HIP events inside the library, wall time including the 2.8 GB upload."""
    recon = pkg.reconstruction
    geo = recon.create_geometry(n, start_angle=90.0)
    u0, v0 = -(nu - 1) / 2 * du, -(nv - 1) / 2 * du
    u = (np.arange(nu, dtype=np.float32) - nu / 2) / nu
    proj = np.empty((n, nv, nu), dtype=np.float32)
    proj[:] = (2.0 * np.sqrt(np.maximum(0.0, 0.16 - u * u)))[None, None, :]  # a cylinder's line integrals (the timing does not depend on the values)
    dim = (464, 250, 464)
    wall, r = None, None
    for rep in range(2):  # the first call pays plan creation and allocations
        t0 = time.perf_counter()
        vol, r = recon.fdk(proj, geo, (du, du), (u0, v0), dim, (1.0, 1.0, 1.0), hann=1.0, hann_y=1.0, pad=pad, gpu_id=device)
        wall = time.perf_counter() - t0
    upd = n * dim[0] * dim[1] * dim[2]
    return {"projections": n, "detector": f"{nu}x{nv}", "volume": "464x250x464", "pad": pad, "ms_filter": r["ms_filter"], "ms_backproject": r["ms_backproject"],
            "ms_kernels": r["ms_filter"] + r["ms_backproject"], "voxel_updates_per_s": upd / (r["ms_backproject"] * 1e-3),
            "wall_s_including_host_transfers": wall, "finite": bool(np.isfinite(vol).all()), "parity": "unpinned against RTK (DESIGN.md 2)"}


def other_workloads(eng, torch, H, projections, device, ceilings=None):
    """Configs 3-5 under the driver's clock: the same kernel measurement (8 launches of H histories) on the bundled CIRS
    phantom and on the patient-like thorax."""
    out = {}
    for wl in ("cirs", "thorax"):
        t0 = time.perf_counter()
        wd = Path(os.path.join(tempfile.gettempdir(), f"mcgpu_bench_{wl}_512_{projections}"))
        inp = wd / "input.in"
        if not (inp.exists() and (wd / "geometry.voxbin").exists()):
            wd.mkdir(parents=True, exist_ok=True)
            build_workload(wd, wl, H, projections, eng)
        t1 = time.perf_counter()
        with eng.create(inp, device=device) as c2:
            k_ms, k_min, detected = timed_launches(c2, torch, H)
            roof, valu = roofline_block(wl, H, k_ms, ceilings, c2)
            out[wl] = {"value": H / (k_ms * 1e-3), "unit": "histories/s", "kernel_ms_avg": k_ms, "kernel_ms_min": k_min, "launches": 8,
                       "config": WORKLOADS[wl][0], "roofline": {k: roof[k] for k in ("kernel", "frac", "achieved", "traffic", "hbm_counter_frac", "fabric_bytes_per_history",
                                                                                     "l2_hit_rate", "traffic_source", "algorithmic_bytes_per_history")},
                       "valu_issue": valu, "volume_bytes": c2.geti("volume_bytes_device"), "materials_used": c2.geti("num_materials_used"),
                       "detected_energy_units_last_projection": detected,
                       # the bit-exact personality on this workload (reference arithmetic, RANECU streams), driver-timed like the rest
                       "compat": {k: v for k, v in compat_leg(c2, torch, H, launches=2).items() if k != "what"},
                       "prepare_inputs_s": t1 - t0, "load_measure_s": time.perf_counter() - t1}
            out[wl]["roofline"]["binding"] = roof.get("binding")
            if wl == "thorax":
                out[wl]["known_deviation_entry_face"] = entry_face_deficit(c2)
            if wl == "cirs":
                t4 = time.perf_counter()
                out["cirs_4d"] = cirs_4d_leg(c2, torch, H)
                out["cirs_4d"]["leg_s"] = time.perf_counter() - t4
    return out


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters: this process has not touched
    the GPU and never does), one per GPU, wait, relay rank 0's JSON line.  Any rank failing fails the run."""
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": env0.get("MASTER_PORT", str(free_port())),
                 "BENCH_SPAWNED": "1", "HSA_ENABLE_IPC_MODE_LEGACY": env0.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    line_file = tempfile.TemporaryFile()  # rank 0's stdout: the one JSON line
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=line_file if r == 0 else sys.stderr.fileno()))
    # wait for all of them; the first rank that fails ends the run (its peers would otherwise sit in a collective until the
    # process group's own timeout)
    deadline = time.time() + float(os.environ.get("BENCH_RANK_TIMEOUT_S", "900"))
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed_now = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if failed_now or time.time() > deadline:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()  # exactly the processes started here
                    codes[r] = p.wait()
                    if not failed_now:
                        print(f"bench.py: rank {r} did not finish in time", file=sys.stderr)
            break
        time.sleep(0.05)
    rc = 0
    for r, code in enumerate(codes):
        if code != 0:
            print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
            rc = rc or (code if code and code > 0 else 1)
    line_file.seek(0)
    line = line_file.read()
    if rc == 0 and not line.strip():
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--histories", type=float, default=1e8, help="histories per projection per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="catphan")
    ap.add_argument("--voxels", type=int, default=512, help="catphan workload: cube edge")
    ap.add_argument("--projections", type=int, default=894)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="host time budget of the cpu_baseline leg (bounded sample of the same workload)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the pipelined-scan measurements after the timed region")
    ap.add_argument("--no-compat", action="store_true", help="skip the COMPAT-personality leg")
    ap.add_argument("--no-workloads", action="store_true", help="skip the CIRS / thorax legs (configs 3-5)")
    ap.add_argument("--no-fdk", action="store_true", help="skip the FDK reconstruction leg (config 4)")
    ap.add_argument("--scan-projections", type=int, default=894, help="projections of the end_to_end leg")
    ap.add_argument("--ascii-projections", type=int, default=64, help="projections of the end_to_end_ascii leg (63 MB of text each)")
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args()
    if args.gpus is None:  # `torchrun ... bench.py` without --gpus: the launcher's rank count is the GPU count
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    # No launcher gave this process a rank: it becomes the launcher (nothing GPU-related has been imported yet).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner from C stdio on
    # its first collective), so everything else is sent to stderr and the line goes to the original descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the rank count is a fact of the launcher; a line that claims another n_gpus would be mislabelled
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to run a mislabelled measurement")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if world > n_dev and not share:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs, this node shows {n_dev} (BENCH_SHARE_GPU=1 lets ranks share a GPU: development only)")
    device = local_rank % n_dev if share else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share and world > n_dev:  # RCCL refuses two ranks on one device: the control plane falls back to gloo
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    backend = dist.get_backend() if dist else None
    # control plane of the fallback decisions: a gloo group beside the RCCL world, so that the ranks can still agree on another
    # route after an RCCL collective has failed (the verdicts below are a few bytes of host data)
    ctl = dist.new_group(backend="gloo") if (dist and backend == "nccl") else None

    def barrier():
        if dist:
            dist.barrier()

    def agree(ok_here: bool) -> bool:
        """True iff `ok_here` is true on EVERY rank (all ranks get the same answer)."""
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok_here), group=ctl)
        return all(flags)

    import cases
    eng = cases.pkg.engine
    eng.load_library()

    H = int(args.histories)
    label, algo_bytes, algo_src = WORKLOADS[args.workload]
    label = label.format(v=args.voxels)
    workdir = Path(args.workdir or os.path.join(tempfile.gettempdir(), f"mcgpu_bench_{args.workload}_{args.voxels}_{args.projections}"))
    inp = workdir / "input.in"
    t_prep0 = time.perf_counter()
    if rank == 0 and not (inp.exists() and (workdir / "geometry.vox").exists() and (workdir / "geometry.voxbin").exists()):
        workdir.mkdir(parents=True, exist_ok=True)
        build_workload(workdir, args.workload, H, args.projections, eng, args.voxels)
    barrier()
    t_prep = time.perf_counter() - t_prep0
    t_load0 = time.perf_counter()
    ctx = eng.create(inp, device=device)
    t_load = time.perf_counter() - t_load0

    nz, nx = ctx.detector_shape
    stream = torch.cuda.current_stream().cuda_stream
    nproj = ctx.num_projections
    seed = ctx.geti("seed")
    kernel_ms = []
    # ---- N > 1: the sum of the per-rank tallies (the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019)
    #   "copy" (default): the engine's tally exchange -- every projection has an owner rank, the others push their tally into
    #     its landing buffer with a copy engine while the next projection is tracked, the owner adds them behind its next
    #     kernel (exchange.cpp; the path the drop-in executable runs between its devices)
    #   "rccl": sharding.reduce_image, one collective per G projections between two tracking kernels (exposed by design)
    #   "none": PROJECTION sharding (SURVEY 8e's fallback): rank r simulates all H histories of its own projections; no exchange,
    #     no collective on the data path.  Not north_star's split (a projection's histories stay on one GPU): the last line of
    #     defence on a node where neither the exchange nor RCCL works.  Fallback order: copy -> rccl -> none, agreed by all ranks.
    exchange_kind = os.environ.get("BENCH_EXCHANGE", "copy") if dist else None
    if exchange_kind not in (None, "copy", "rccl", "none"):
        raise SystemExit(f"bench.py: BENCH_EXCHANGE={exchange_kind}: expected copy, rccl or none")
    if dist and backend == "gloo" and exchange_kind == "rccl":
        raise SystemExit("bench.py: BENCH_EXCHANGE=rccl needs one GPU per rank")
    fallbacks = []  # routes tried and given up, with the reason (reported in config.parallelism_fallbacks)
    x = shared_map = None
    policy = eng.EXCHANGE_ROTATE if os.environ.get("MCGPU_EXCHANGE_POLICY", "1") != "0" else eng.EXCHANGE_ROOT0
    if exchange_kind == "copy":
        shm = Path("/dev/shm") / f"mcgpu_exchange_{os.environ['MASTER_PORT']}"
        if rank == 0:
            shared_map = eng.Exchange.open_shared(shm, world, create=True)
        barrier()
        if rank != 0:
            shared_map = eng.Exchange.open_shared(shm, world, create=False)
        try:
            x = eng.Exchange(device, rank, world, ctx.image_words, shared_map, policy)
            why = None
        except eng.EngineError as e:
            x, why = None, e
        ok, err = cases.pkg.sharding.connect_exchange(x, dist, group=ctl)  # the same verdict on every rank
        if rank == 0:
            shm.unlink(missing_ok=True)  # every rank holds its mapping
        first_step = 0
        if ok:
            # dry run of the whole protocol on empty tallies, both buffer parities: copy-engine pushes into IPC memory of ANOTHER
            # device, stream waits on interprocess events, the fused add -- everything a real step does except the tracking kernel.
            # A node on which any of that fails between two devices falls back to RCCL with all ranks, here, not in the timed region.
            try:
                for k in (0, 1):
                    x.begin(k, stream)
                    x.submit(k, stream)
                for k in (0, 1):
                    x.collect(k, stream)
                torch.cuda.synchronize()
                dry = None
            except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
                dry = e
            ok, err = agree(dry is None), (dry or err)
            first_step = 2
        if not ok:
            # no IPC between these ranks' devices (or the runtime refused an interprocess event): every rank falls back to the
            # RCCL reduction together -- slower (the collective is exposed between kernels), but a measurement instead of a failure
            nxt = "none" if backend == "gloo" else "rccl"  # ranks that share a GPU have no RCCL to fall back to
            print(f"bench.py: rank {rank}: the tally exchange is not available here ({why or err}); falling back to BENCH_EXCHANGE={nxt}", file=sys.stderr)
            fallbacks.append({"route": "copy", "reason": str(why or err)[:300]})
            if x:
                x.close()
            x = None
            exchange_kind = nxt
    G = max(1, int(os.environ.get("BENCH_REDUCE_GROUP", "8"))) if exchange_kind == "rccl" else 1
    images = None if x else torch.zeros((G, 4, nz, nx), dtype=torch.int64, device="cuda")
    filled = [0]
    narrow = exchange_kind == "rccl" and os.environ.get("BENCH_REDUCE_U32", "1") == "1"
    reduce_algo = os.environ.get("BENCH_REDUCE_ALGO", "scatter")
    reduce_bytes = [0]
    last_reduced = [None]  # device pointer / tensor of the last complete tally this rank holds
    n_step = [first_step if x else 0]  # exchange step counter (consecutive over the dry run, warm-up, timed region and the check)

    def reduce_group():
        if exchange_kind == "rccl" and filled[0] > 0:
            # on the current stream: the next tracking launch waits for it
            reduce_bytes[0] += cases.pkg.sharding.reduce_image(images[:filled[0]], dst=0, narrow=narrow, algorithm=reduce_algo)
        filled[0] = 0

    collected = [first_step if x else 0]  # exchange steps collected so far (each step is collected exactly once, in order)

    def collect_up_to(k_excl):
        while x and collected[0] < k_excl:
            got = x.collect(collected[0], stream)
            last_reduced[0] = got or last_reduced[0]
            collected[0] += 1

    kernel_events = []     # (start, stop) HIP events around every timed launch, on the stream it is launched on; read after the region

    def step(i, timed, hist=H, projection=None):
        # spread the sampled projections over the arc; projection sharding: step i of rank r is projection number i * world + r
        # of that sequence (sharding.shard_projections), simulated whole by this rank
        k_seq = i * world + rank if exchange_kind == "none" else i
        p = (k_seq * 149) % nproj if projection is None else projection
        first_id = 0 if exchange_kind == "none" else rank * hist  # history sharding: disjoint history ids per rank
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
        if x:
            k = n_step[0]
            tally = x.begin(k, stream)
            if ev:
                ev[0].record()
            ctx.launch(p, tally, hist, mode="fast", seed=seed, first=first_id, stream=stream)
            if ev:
                ev[1].record()
            x.submit(k, stream)
            collect_up_to(k)  # step k - 1, behind this kernel: its pushes had the whole kernel to land
            n_step[0] = k + 1
        else:
            image = images[filled[0]]
            ctx.clear(image.data_ptr(), stream)
            if ev:
                ev[0].record()
            ctx.launch(p, image.data_ptr(), hist, mode="fast", seed=seed, first=first_id, stream=stream)
            if ev:
                ev[1].record()
            filled[0] += 1
            last_reduced[0] = image
        if ev:
            kernel_events.append(ev)  # no host wait inside the timed region: the stream never runs dry between two projections
        if not x and filled[0] == G:
            reduce_group()

    def drain():
        collect_up_to(n_step[0])
        reduce_group()

    if exchange_kind == "rccl":
        # the reductions of the timed region (full groups of G projections and the remainder group) run once untimed: RCCL
        # sets up its channels and sharding.reduce_image its staging buffers on the first call with a payload shape.  A node
        # on which that fails makes all ranks take projection sharding together (agreed over the gloo control group).
        try:
            for size in sorted({G if args.steps >= G else 0, args.steps % G, G if args.warmup >= G else 0} - {0}):
                cases.pkg.sharding.reduce_image(images[:size], dst=0, narrow=narrow, algorithm=reduce_algo)
            torch.cuda.synchronize()
            trial = None
        except Exception as e:  # noqa: BLE001 -- reported, and agreed on below
            trial = e
        if not agree(trial is None):
            print(f"bench.py: rank {rank}: the RCCL reduction failed here ({trial}); falling back to projection sharding (BENCH_EXCHANGE=none)", file=sys.stderr)
            fallbacks.append({"route": "rccl", "reason": str(trial)[:300]})
            exchange_kind = "none"
    for i in range(args.warmup):
        step(i, False)
    drain()
    barrier()
    torch.cuda.synchronize()
    reduce_bytes[0] = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    drain()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms.extend(a.elapsed_time(b) for a, b in kernel_events)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: correctness of the sharded sum, and what the exchange cost
    multi = None
    if dist:
        h = int(float(os.environ.get("BENCH_CHECK_HISTORIES", "1e7")))
        p_chk = 447 % nproj
        if x:
            k_chk = n_step[0]
            step(0, False, hist=h, projection=p_chk)
            drain()
            owner = x.owner(k_chk)
            if rank == owner:
                sharded = ctx.download_image(last_reduced[0], stream)
        elif exchange_kind == "none":
            # projection sharding: every rank simulates a projection of its own, whole; rank 0 then repeats each of them alone
            # and compares the words (all ranks run the same code on the same inputs: this checks the plumbing, e.g. that no
            # rank's tally leaked into another's)
            owner = 0
            filled[0] = 0
            step(0, False, hist=h, projection=(p_chk + rank) % nproj)
            torch.cuda.synchronize()
            mine = hashlib.sha256(images[0].cpu().numpy().tobytes()).hexdigest()
            digests = [None] * world
            dist.all_gather_object(digests, mine, group=ctl)
        else:
            owner = 0
            step(0, False, hist=h, projection=p_chk)
            drain()
            if rank == 0:
                torch.cuda.synchronize()
                sharded = images[0].cpu().numpy().view(np.uint64)
        barrier()
        verdict = None
        if exchange_kind == "none":
            if rank == 0:
                bad = 0
                for r_ in range(world):
                    alone, _, _ = ctx.run_projection((p_chk + r_) % nproj, h, mode="fast", seed=seed, first=0)
                    bad += int(hashlib.sha256(np.ascontiguousarray(alone).view(np.int64).tobytes()).hexdigest() != digests[r_])
                verdict = {"passed": bad == 0, "what": "every rank's own projection equals rank 0's run of that projection (SHA-256 of the tally)",
                           "projections": [(p_chk + r_) % nproj for r_ in range(world)], "histories": h, "ranks": world, "ranks_differing": bad}
        elif rank == owner:  # the others idle: one rank simulates ALL the history ids [0, world * h) of that projection alone
            alone, _, done = ctx.run_projection(p_chk, world * h, mode="fast", seed=seed, first=0)
            verdict = {"passed": bool(np.array_equal(alone, sharded)), "projection": p_chk, "histories_per_rank": h, "ranks": world,
                       "checked_on_rank": rank, "words_differing": int(np.count_nonzero(alone != sharded)),
                       "detected_energy_units": int(alone.sum())}
        verdicts = [None] * world
        dist.all_gather_object(verdicts, verdict, group=ctl)
        stats = [None] * world
        dist.all_gather_object(stats, (x.stats() if x else None, float(np.mean(kernel_ms))), group=ctl)
        multi = {"sharded_equals_single": verdicts[owner], "stats": stats}

    detected = 0
    if last_reduced[0] is not None and (not dist):
        detected = int(last_reduced[0].sum().item())

    failed = False
    if rank == 0:
        total_hist = float(H) * world * args.steps
        value = total_hist / elapsed
        k_ms = float(np.mean(kernel_ms))
        ceilings = measured_ceilings(ctx) if world == 1 else None
        roof, valu = roofline_block(args.workload, H, k_ms, ceilings, ctx)
        out = {
            "metric": "photon histories/sec (512^3 vol, 894 proj)", "value": value, "unit": "histories/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}_{args.projections}proj_{H:.0e}hist_per_proj_per_gpu",
                       "detector": f"{nx}x{nz}", "histories_per_projection_per_gpu": H, "kernel": "fast",
                       "parallelism": (f"PROJECTION-sharded x{world}: every rank simulates whole projections, no exchange and no collective (SURVEY 8e fallback mode; NOT north_star's history split)"
                                       if exchange_kind == "none" else
                                       f"history-sharded x{world}" + ("" if not dist else
                                      (", tally exchange: copy-engine pushes to the projection's owner (" + ("owner = projection mod ranks" if policy == eng.EXCHANGE_ROTATE else "owner = rank 0") + "), one fused add per projection"
                                       if x else f", one RCCL sum-reduction ({reduce_algo}) of the detector tallies per {G} projections"))),
                       "parallelism_route": exchange_kind, "parallelism_fallbacks": fallbacks if dist else None,
                       "ranks_started_by": "bench.py itself" if os.environ.get("BENCH_SPAWNED") else ("an external launcher" if dist else "single process"),
                       "process_group_backend": backend, "ranks_share_gpus": bool(share and world > n_dev),
                       "volume_kind": ["u8-palette", "u16-palette", "raw-float2"][ctx.geti("volume_kind")],
                       "volume_bytes": ctx.geti("volume_bytes_device"), "materials_used": ctx.geti("num_materials_used"),
                       "lds_bytes_per_workgroup": ctx.geti("lds_bytes_fast"), "workgroups_per_cu": ctx.geti("blocks_per_cu"),
                       "second_level": "tile records" if ctx.geti("tile_records") else "none", "tiles_in_mixed_bricks": ctx.geti("tiles_in_mixed_bricks"),
                       "per_gpu_value": value / world},
            # frac: the reference algorithm's bytes per history (what a history NEEDS in the reference layout) over the kernel
            # time, against the HBM peak -- a model figure.  hbm_counter_frac: the bytes that actually crossed the HBM interface
            # (PMC counters of this kernel build) over the same time.
            "roofline": roof,
            "valu_issue": valu,
            "measured_ceilings": ceilings,
            "timing": {"prepare_inputs_s": t_prep, "load_and_upload_s": t_load},
            "check": {"detected_energy_units_last_projection": detected},
        }
        if multi:
            v = multi["sharded_equals_single"]
            out["check"]["projection_sharded_equals_single" if exchange_kind == "none" else "sharded_equals_single"] = v
            out["check"]["passed"] = bool(v and v["passed"])
            failed = failed or not out["check"]["passed"]
            k_all = [s_[1] for s_ in multi["stats"]]
            red = {"kind": exchange_kind, "kernel_ms_avg_per_rank": k_all, "step_minus_slowest_kernel_ms": elapsed / args.steps * 1e3 - max(k_all)}
            if x:
                st = [s_[0] for s_ in multi["stats"]]
                push = [a["last_push_ms"] for a in st if a["pushes"] > 0]
                add = [a["last_add_ms"] for a in st if a["collects"] > 0 and a["last_add_ms"] > 0]
                bytes_push = st[0]["bytes_per_push"]
                red.update({"bytes_per_push": bytes_push, "pushes_per_projection": world - 1,
                            # HIP events around the copy on the copy stream.  Without a profiler attached they bracket the
                            # SUBMISSION of a copy-engine transfer, not its duration, on some runs (a 45 MB push cannot take less than
                            # 0.7 ms at the engine's 60 GB/s): such a reading is flagged instead of being turned into a bandwidth;
                            # the profiler's figure is in profiles/r03i_exchange_overlap_rocprofv3_memory_copy_stats.txt
                            "push_ms_by_events": float(np.max(push)) if push else None,
                            "push_GBps": (bytes_push / (float(np.max(push)) * 1e-3) / 1e9 if push and bytes_push / (float(np.max(push)) * 1e-3) / 1e9 < 100.0 else None),
                            "push_events_bracket_submission_only": bool(push and bytes_push / (float(np.max(push)) * 1e-3) / 1e9 >= 100.0),
                            "fused_add_ms": float(np.max(add)) if add else None,
                            # what a tracking stream sees of the exchange per projection it OWNS: the fused add (+ a 45 MB memset per
                            # step on every rank, inside begin(), which the N = 1 step pays as well)
                            "exposed_ms_per_step_on_the_critical_rank": (float(np.max(add)) if add else 0.0) * (1.0 / world if policy == eng.EXCHANGE_ROTATE else 1.0),
                            "host_wait_s_per_rank": [a["host_wait_s"] for a in st],
                            "owner_policy": "rotate" if policy == eng.EXCHANGE_ROTATE else "rank0"})
            elif exchange_kind == "rccl":
                red.update({"bytes_per_rank_in_timed_region": reduce_bytes[0], "narrowed_to_u32_when_it_fits": bool(narrow), "algorithm": reduce_algo,
                            "projections_per_reduction": G})
            out["reduce"] = red
        if world == 1:
            if not args.no_compat:
                out["compat"] = compat_leg(ctx, torch, H)
                out["check"]["fast_vs_compat"] = fast_vs_compat_check(ctx)
                failed = failed or not out["check"]["fast_vs_compat"]["passed"]
            if not args.no_end_to_end:
                out["end_to_end"] = end_to_end_scan(ctx, H, workdir, n=min(args.scan_projections, nproj))
                if args.ascii_projections > 0:
                    out["end_to_end_ascii"] = end_to_end_scan(ctx, H, workdir, n=min(args.ascii_projections, nproj), ascii_files=True)
            if not args.no_workloads and args.workload == "catphan":
                out["workloads"] = other_workloads(eng, torch, H, args.projections, device, ceilings)
                kd = out["workloads"]["thorax"].get("known_deviation_entry_face")
                out["check"]["known_deviations"] = {"beam_edge_column": (out["check"].get("fast_vs_compat") or {}).get("beam_edge_column"), "entry_face": kd}
                failed = failed or not (kd is None or kd["passed"])
                if not args.no_fdk:
                    out["fdk"] = fdk_leg(cases.pkg, device)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only (the other ranks would idle meanwhile)
            base, img_cpu, w2_cpu, n_cpu = cpu_baseline(ctx, seconds_budget=args.cpu_seconds)
            out["cpu_baseline"] = base
            # a second ceiling: one scattered 64-bit atomic add per detected photon; rate measured by
            # tools/archive/micro/atomic_rate.hip on MI355X = 2.37e10/s
            tally_hits = base["events_per_history"]["tally_hits"]
            a_peak = ceilings["scattered_64bit_atomic_adds_per_s"] / 1e9 if ceilings else 23.7  # 23.7: tools/archive/micro/atomic_rate.hip, round 2
            out["atomic_roofline"] = {"bound": "scattered 64-bit atomic adds", "achieved": tally_hits * H / (k_ms * 1e-3) / 1e9, "peak": a_peak,
                                      "peak_source": "measured in this run (mcgpu_microbench)" if ceilings else "tools/archive/micro/atomic_rate.hip, round 2",
                                      "unit": "Gatomic/s", "frac": tally_hits * H / (k_ms * 1e-3) / (a_peak * 1e9),
                                      "detected_photons_per_history": tally_hits}
            out["check"].update(oracle_check(ctx, H, img_cpu, w2_cpu, n_cpu))
            out["check"]["passed"] = bool(out["check"]["passed"] and out["check"].get("fast_vs_compat", {}).get("passed", True))
            failed = failed or not out["check"]["passed"]
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    barrier()
    if x:
        torch.cuda.synchronize()
        barrier()  # nobody unmaps landing memory a peer may still address
        x.close()
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: the correctness check FAILED (see `check` in the line above)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
