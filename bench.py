#!/usr/bin/env python3
"""bench.py -- photon histories/s of the MC CBCT projection hot path on MI355X.

Workload (BASELINE.json configs[1], `--workload catphan`, the default): Catphan604 phantom in a 512^3 volume @ 1 mm, Varian
half-fan geometry, 1848x768 detector, 894-projection trajectory, 1e8 histories per projection per GPU, default spectrum, real
PENELOPE material tables.  `--workload cirs` / `thorax` run the same measurement on the bundled CIRS phantom (configs 3/5
geometry) and on the patient-like 512x512x256 thorax (config 4 shape); they are not the headline.

A "step" is one projection: the photon-history kernel over one batch of histories (FAST personality), plus -- for N > 1 -- the
RCCL sum-reduce of the detector tally to rank 0.  Inputs (volume, tables) are resident in HBM before the timed region.
Weak scaling: every rank simulates `--histories` histories of each projection with its own disjoint history-id range.

Prints ONE JSON line on rank 0 (driver contract) carrying `roofline` and `cpu_baseline`, plus (1 GPU) a driver-timed leg of the
COMPAT personality (`compat`: the reference's RANECU streams and arithmetic, bit-identical to the oracle), a correctness
figure tied to the oracle (`check`) and the pipelined-scan figure (`end_to_end`).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
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
    """COMPAT personality (RANECU leap-frog streams, the reference's arithmetic, bit-identical to the oracle) timed like the
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
            "what": "COMPAT kernel: RANECU streams + reference arithmetic, tallies bit-identical to the CPU oracle (tests/test_gpu_fullsize.py)"}


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


def end_to_end_scan(ctx, H, workdir, n=224):
    """The pipelined scan driver (track -> finalize -> pinned copy -> writer thread) with the three MetaImage stacks
    written to disk: per-projection wall time including output, reported beside the kernel-only figure."""
    out = workdir / "scan_out"
    out.mkdir(exist_ok=True)
    crop = 1024 if ctx.detector_shape[1] == 1848 else 0
    rep = ctx.run_scan(mode="fast", first_projection=100, num_projections=n, histories=H, crop_nx=crop, write_stacks=True, output_folder=out,
                       pixel_spacing=(0.776, 0.776))
    for f in out.glob("projections_*.mha"):
        f.unlink()
    return {"projections": n, "seconds_total": rep["seconds_total"], "histories_per_s_with_stacks": n * H / rep["seconds_total"],
            "ms_per_projection_with_stacks": rep["seconds_total"] / n * 1e3, "ms_per_projection_kernels": rep["seconds_kernels"] / n * 1e3,
            "writer_ms_per_projection": rep["seconds_writer"] / n * 1e3, "drain_after_last_kernel_ms": rep["seconds_after_last_kernel"] * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--histories", type=float, default=1e8, help="histories per projection per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="catphan")
    ap.add_argument("--voxels", type=int, default=512, help="catphan workload: cube edge")
    ap.add_argument("--projections", type=int, default=894)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the pipelined-scan measurement after the timed region")
    ap.add_argument("--no-compat", action="store_true", help="skip the COMPAT-personality leg")
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner from C stdio on
    # its first collective), so everything else is sent to stderr and the line goes to the original descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # the latter: exercise the collective path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

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
    if dist:
        dist.barrier()
    t_prep = time.perf_counter() - t_prep0
    t_load0 = time.perf_counter()
    ctx = eng.create(inp, device=local_rank)
    t_load = time.perf_counter() - t_load0

    nz, nx = ctx.detector_shape
    # N > 1: every rank tracks its history shard of G consecutive projections into G tally buffers, then ONE sum-reduction
    # brings the G tallies to rank 0 (the reference's per-projection MPI_Reduce, MC-GPU_v1.3.cu:1019, batched: fewer, larger
    # messages; sharding.reduce_image: slices straight to their owners over the point-to-point xGMI links, summed there,
    # gathered on rank 0).  The reduce is ordered between two tracking kernels on purpose: a kernel that is
    # still running while the persistent tracking grid is dispatched fragments the CUs' register files for the whole
    # launch and costs up to 30 % (tools/placement_probe.py), so nothing overlaps a tracking launch.
    # Payload: 32-bit words whenever the summed tallies provably fit (sharding.reduce_image), else 64-bit.
    G = max(1, int(os.environ.get("BENCH_REDUCE_GROUP", "8"))) if dist else 1
    images = torch.zeros((G, 4, nz, nx), dtype=torch.int64, device="cuda")
    filled = [0]
    stream = torch.cuda.current_stream().cuda_stream
    nproj = ctx.num_projections
    seed = ctx.geti("seed")
    kernel_ms = []
    narrow = dist is not None and os.environ.get("BENCH_REDUCE_U32", "1") == "1"
    reduce_algo = os.environ.get("BENCH_REDUCE_ALGO", "scatter")  # sharding.reduce_image: slices to their owners, sum, gather
    reduce_bytes = [0]

    def reduce_group():
        if dist and filled[0] > 0:
            # on the current stream: the next tracking launch waits for it (see above)
            reduce_bytes[0] += cases.pkg.sharding.reduce_image(images[:filled[0]], dst=0, narrow=narrow, algorithm=reduce_algo)
        filled[0] = 0

    def step(i, timed):
        p = (i * 149) % nproj  # spread the sampled projections over the arc
        image = images[filled[0]]
        ctx.clear(image.data_ptr(), stream)
        # disjoint history ids per rank: [rank*H, (rank+1)*H)
        ctx.launch(p, image.data_ptr(), H, mode="fast", seed=seed, first=rank * H, stream=stream)
        filled[0] += 1
        last[0] = filled[0] - 1
        if timed:
            kernel_ms.append(ctx.last_kernel_ms())  # waits for this launch only
        if filled[0] == G:
            reduce_group()

    last = [0]
    drain = reduce_group

    for i in range(args.warmup):
        step(i, False)
    drain()
    if dist:
        # the reductions of the timed region (full groups of G projections and the remainder group) run once untimed: RCCL
        # sets up its channels and sharding.reduce_image its staging buffers on the first call with a payload shape
        for size in sorted({G if args.steps >= G else 0, args.steps % G} - {0}):
            cases.pkg.sharding.reduce_image(images[:size], dst=0, narrow=narrow, algorithm=reduce_algo)
        dist.barrier()
    torch.cuda.synchronize()
    reduce_bytes[0] = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    detected = int(images[last[0]].sum().item()) if rank == 0 else 0

    if rank == 0:
        total_hist = float(H) * world * args.steps
        value = total_hist / elapsed
        k_ms = float(np.mean(kernel_ms))
        achieved = algo_bytes * H / (k_ms * 1e-3) / 1e9
        pmc, pmc_src = pmc_summary(args.workload) if H == int(1e8) else (None, "summary is per 1e8-history launch")
        traffic = hbm_counter_frac = valu = None
        if pmc:
            # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled
            # (MI355X_MICROARCH.md, HBM section)
            traffic = (2.0 * pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
            hbm_counter_frac = traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            insts = pmc["SQ_INSTS_VALU"]["mean_per_dispatch"]
            # ceiling: a dense dependent-FMA kernel, 8 waves/SIMD, 16-32 active lanes, on the same chip (tools/micro/exec_skip.hip:
            # 5.24e9 wave-instructions on 1024 SIMDs in 5.1 ms)
            peak = 5.24e9 / 1024.0 / 5.1e6
            valu = {"valu_wave_instructions_per_launch": insts, "achieved_per_ns_per_simd": insts / 1024.0 / (k_ms * 1e6),
                    "measured_peak_per_ns_per_simd": peak, "frac": insts / 1024.0 / (k_ms * 1e6) / peak,
                    "lane_utilisation": pmc["SQ_THREAD_CYCLES_VALU"]["mean_per_dispatch"] / pmc["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / 64.0}
        out = {
            "metric": "photon histories/sec (512^3 vol, 894 proj)", "value": value, "unit": "histories/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}_{args.projections}proj_{H:.0e}hist_per_proj_per_gpu",
                       "detector": f"{nx}x{nz}", "histories_per_projection_per_gpu": H, "kernel": "fast",
                       "parallelism": f"history-sharded x{world}" + (f", one RCCL sum-reduction ({reduce_algo}) of the detector tallies per {G} projections" if dist else ""),
                       "volume_kind": ["u8-palette", "u16-palette", "raw-float2"][ctx.geti("volume_kind")],
                       "volume_bytes": ctx.geti("volume_bytes_device"), "materials_used": ctx.geti("num_materials_used"),
                       "lds_bytes_per_workgroup": ctx.geti("lds_bytes_fast"), "workgroups_per_cu": ctx.geti("blocks_per_cu"),
                       "per_gpu_value": value / world},
            # frac: the reference algorithm's bytes per history (what a history NEEDS in the reference layout) over the kernel
            # time, against the HBM peak -- a model figure.  hbm_counter_frac: the bytes that actually crossed the HBM interface
            # (PMC counters of this kernel build) over the same time.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "hbm_counter_frac": hbm_counter_frac, "traffic_source": pmc_src,
                         "kernel": "track_pool_kernel<u8> (fast)", "kernel_ms_avg": k_ms, "kernel_source_sha16": kernel_source_hash(),
                         "algorithmic_bytes_per_history": algo_bytes, "algorithmic_bytes_source": algo_src,
                         "algorithmic_bytes_per_launch": algo_bytes * H},
            "valu_issue": valu,
            "timing": {"prepare_inputs_s": t_prep, "load_and_upload_s": t_load},
            "check": {"detected_energy_units_last_projection": detected},
        }
        if dist:
            out["reduce"] = {"bytes_per_rank_in_timed_region": reduce_bytes[0], "narrowed_to_u32_when_it_fits": bool(narrow), "algorithm": reduce_algo}
        if world == 1:
            if not args.no_compat:
                out["compat"] = compat_leg(ctx, torch, H)
            if not args.no_end_to_end:
                out["end_to_end"] = end_to_end_scan(ctx, H, workdir)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only (the other ranks would idle meanwhile)
            base, img_cpu, w2_cpu, n_cpu = cpu_baseline(ctx)
            out["cpu_baseline"] = base
            # a second ceiling: one scattered 64-bit atomic add per detected photon; rate measured by
            # tools/micro/atomic_rate.hip on MI355X = 2.37e10/s
            tally_hits = base["events_per_history"]["tally_hits"]
            out["atomic_roofline"] = {"bound": "scattered 64-bit atomic adds", "achieved": tally_hits * H / (k_ms * 1e-3) / 1e9, "peak": 23.7,
                                      "unit": "Gatomic/s", "frac": tally_hits * H / (k_ms * 1e-3) / 23.7e9,
                                      "detected_photons_per_history": tally_hits}
            if world == 1:
                out["check"].update(oracle_check(ctx, H, img_cpu, w2_cpu, n_cpu))
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
