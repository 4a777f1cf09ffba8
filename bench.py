#!/usr/bin/env python3
"""bench.py -- photon histories/s of the MC CBCT projection hot path on MI355X.

Workload (BASELINE.json configs[1]): Catphan604 phantom in a 512^3 volume @ 1 mm, Varian half-fan geometry,
1848x768 detector, 894-projection trajectory, 1e8 histories per projection per GPU, default spectrum,
real PENELOPE material tables.  A "step" is one projection: the photon-history kernel over one batch of
histories (FAST personality), plus -- for N > 1 -- the RCCL sum-reduce of the 45 MB detector tally to
rank 0.  Inputs (volume, tables) are resident in HBM before the timed region.  Weak scaling: every rank
simulates `--histories` histories of each projection with its own disjoint history-id range.

Prints ONE JSON line on rank 0 (driver contract) carrying `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
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

# Algorithmic bytes per history, Catphan604, reference table layout:
# 8 B x 21.26 voxel gathers + 24 B x 1.84 MFP rows + 8 B x 1.47 Woodcock rows + 16 B x 0.93 tally RMW.
ALGO_BYTES_PER_HISTORY = 241.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def build_workload(workdir: Path, n_vox: int, histories: int, n_proj: int, engine):
    """Catphan604 geometry + input file in the reference's wire formats (written once, by rank 0)."""
    import cases
    pkg = cases.pkg
    geo = pkg.geometry.MCCatPhan604Geometry(shape=(n_vox,) * 3, image_spacing=(1.0, 1.0, 1.0))
    mats = cases.material_files()
    sim = pkg.simulation.MCSimulation(geo, mats, cases.spectrum_file(), n_histories=histories, n_projections=n_proj,
                                      angle_between_projections=360.0 / n_proj)
    # geometry.vox (the reference's text format) + geometry.voxbin (binary sidecar the engine prefers: no 134 M-line parse)
    return sim.prepare_simulation(workdir, compress_geometry=False, engine=engine, binary_sidecar=True)


def cpu_baseline(ctx, seconds_budget: float = 14.0):
    """The restated CPU oracle (oracle/mcgpu_oracle.c, LIBM math == reference arithmetic) on a bounded
    sample of the same workload, on this host's cores.  Reported, never the target."""
    import oracle_lib as ol
    import parity
    T = parity.tables_from_context(ctx)
    cores = os.cpu_count() or 1
    hpt = 150
    # one core first (per-core rate), then all cores in chunks of ~3 s until the budget is used
    t0 = time.perf_counter()
    cnt = ol.OracleCounters()
    T.track(0, 42, 0, 400, hpt, ol.MATH_LIBM, n_threads=1, counters=cnt)
    rate1 = 400 * hpt / (time.perf_counter() - t0)
    batch0, done_batches, elapsed = 400, 0, 0.0
    nb = cores * 16
    while elapsed < seconds_budget:
        t0 = time.perf_counter()
        T.track(0, 42, batch0, nb, hpt, ol.MATH_LIBM, n_threads=cores, counters=cnt)
        dt = time.perf_counter() - t0
        batch0 += nb
        if dt > 0.5:  # chunks too short to time OpenMP start-up fairly are warm-up only
            done_batches += nb
            elapsed += dt
        nb = int(max(cores * 16, min(nb * 3.0 / max(dt, 1e-3), 4e6)))
    c = cnt.as_dict()
    h = float(c["histories"])
    per_hist = {k: round(c[k] / h, 4) for k in ("steps", "voxel_reads", "mfp_reads", "woodcock_reads", "compton", "rayleigh", "photo", "rng", "tally_calls", "tally_hits")}
    return {
        "value": done_batches * hpt / elapsed, "unit": "histories/s", "cores": cores, "kind": "port",
        "sample": f"{done_batches * hpt} histories of projection 0 of the same workload in {elapsed:.1f} s, OpenMP over RANECU batches (oracle/mcgpu_oracle.c, libm math)",
        "per_core_value": rate1, "events_per_history": per_hist,
    }


def pmc_traffic(kernel_ms: float):
    """HBM-side bytes per launch from the committed rocprofv3 PMC summary of this kernel (separate --pmc passes,
    tools/pmc_collect.sh): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request,
    so it is doubled (MI355X_MICROARCH.md, HBM section).  None when no summary is committed."""
    f = ROOT / "profiles" / "pmc_summary_latest.json"
    if not f.exists():
        return None, None
    d = json.loads(f.read_text())
    try:
        fetch, write = d["FETCH_SIZE"]["mean_per_dispatch"], d["WRITE_SIZE"]["mean_per_dispatch"]
    except KeyError:
        return None, None
    return (2.0 * fetch + write) * 1024.0, d.get("_note", f.name)


def valu_issue(kernel_ms: float):
    """VALU issue rate of the tracking kernel against the rate a dense dependent-FMA kernel reaches on the same chip
    (tools/micro/exec_skip.hip: 5.24e9 wave-instructions on 1024 SIMDs in 6.39 ms with 64 active lanes, in 4.96-5.31 ms with
    16-32 active lanes; the tracking kernel runs at 36 % lane utilisation).  SQ_INSTS_VALU from the committed PMC summary."""
    f = ROOT / "profiles" / "pmc_summary_latest.json"
    if not f.exists():
        return None
    d = json.loads(f.read_text())
    if "SQ_INSTS_VALU" not in d:
        return None
    insts = d["SQ_INSTS_VALU"]["mean_per_dispatch"]
    achieved = insts / 1024.0 / (kernel_ms * 1e6)  # wave-instructions per ns and SIMD
    peak = 5.24e9 / 1024.0 / 5.1e6               # measured, 16-32 active lanes
    return {"valu_wave_instructions_per_launch": insts, "achieved_per_ns_per_simd": achieved, "measured_peak_per_ns_per_simd": peak,
            "frac": achieved / peak, "lane_utilisation": d["SQ_THREAD_CYCLES_VALU"]["mean_per_dispatch"] / d["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / 64.0
            if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d else None}


def end_to_end_scan(ctx, H, workdir, n=12):
    """The pipelined scan driver (track -> finalize -> pinned copy -> writer thread) with the three MetaImage stacks
    written to disk: per-projection wall time including output, reported beside the kernel-only figure."""
    out = workdir / "scan_out"
    out.mkdir(exist_ok=True)
    rep = ctx.run_scan(mode="fast", first_projection=100, num_projections=n, histories=H, crop_nx=1024, write_stacks=True, output_folder=out,
                       pixel_spacing=(0.776, 0.776))
    for f in out.glob("projections_*.mha"):
        f.unlink()
    return {"projections": n, "ms_per_projection_with_stacks": rep["seconds_total"] / n * 1e3, "ms_per_projection_kernels": rep["seconds_kernels"] / n * 1e3,
            "writer_ms_per_projection": rep["seconds_writer"] / n * 1e3, "drain_after_last_kernel_ms": rep["seconds_after_last_kernel"] * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--histories", type=float, default=1e8, help="histories per projection per GPU")
    ap.add_argument("--voxels", type=int, default=512)
    ap.add_argument("--projections", type=int, default=894)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the pipelined-scan measurement after the timed region")
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
    workdir = Path(args.workdir or os.path.join(tempfile.gettempdir(), f"mcgpu_bench_{args.voxels}_{args.projections}"))
    inp = workdir / "input.in"
    t_prep0 = time.perf_counter()
    if rank == 0 and not (inp.exists() and (workdir / "geometry.vox").exists() and (workdir / "geometry.voxbin").exists()):
        workdir.mkdir(parents=True, exist_ok=True)
        build_workload(workdir, args.voxels, H, args.projections, eng)
    if dist:
        dist.barrier()
    t_prep = time.perf_counter() - t_prep0
    t_load0 = time.perf_counter()
    ctx = eng.create(inp, device=local_rank)
    t_load = time.perf_counter() - t_load0

    nz, nx = ctx.detector_shape
    # N > 1: every rank tracks its history shard of G consecutive projections into G tally buffers, then ONE RCCL
    # sum-reduce brings the G tallies to rank 0 (the reference's per-projection MPI_Reduce, MC-GPU_v1.3.cu:1019, batched:
    # fewer, larger messages over xGMI).  The reduce is ordered between two tracking kernels on purpose: a kernel that is
    # still running while the persistent tracking grid is dispatched fragments the CUs' register files for the whole
    # launch and costs up to 30 % (tools/placement_probe.py), so nothing overlaps a tracking launch.
    G = max(1, int(os.environ.get("BENCH_REDUCE_GROUP", "8"))) if dist else 1
    images = torch.zeros((G, 4, nz, nx), dtype=torch.int64, device="cuda")
    filled = [0]
    stream = torch.cuda.current_stream().cuda_stream
    nproj = ctx.num_projections
    seed = ctx.geti("seed")
    kernel_ms = []

    def reduce_group():
        if dist and filled[0] > 0:
            dist.reduce(images[:filled[0]], dst=0, op=dist.ReduceOp.SUM)  # the current stream waits for it
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
        dist.barrier()
    torch.cuda.synchronize()
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
        achieved = ALGO_BYTES_PER_HISTORY * H / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(k_ms)
        out = {
            "metric": "photon histories/sec (512^3 vol, 894 proj)", "value": value, "unit": "histories/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"catphan604_{args.voxels}cube_1mm_{args.projections}proj_{H:.0e}hist_per_proj_per_gpu",
                       "detector": f"{nx}x{nz}", "histories_per_projection_per_gpu": H, "kernel": "fast",
                       "parallelism": f"history-sharded x{world}" + (f", one RCCL sum-reduce of the detector tallies per {G} projections" if dist else ""),
                       "volume_kind": ["u8-palette", "u16-palette", "raw-float2"][ctx.geti("volume_kind")],
                       "volume_bytes": ctx.geti("volume_bytes_device"), "per_gpu_value": value / world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "track_pool_kernel<u8> (fast)", "kernel_ms_avg": k_ms,
                         "algorithmic_bytes_per_history": ALGO_BYTES_PER_HISTORY, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_HISTORY * H},
            # second ceiling: one scattered 64-bit atomic add per detected photon; rate measured by
            # tools/micro/atomic_rate.hip on MI355X = 2.37e10/s; detected photons per history of this workload = 0.754
            "atomic_roofline": {"bound": "scattered 64-bit atomic adds", "achieved": 0.754 * H / (k_ms * 1e-3) / 1e9, "peak": 23.7,
                                "unit": "Gatomic/s", "frac": 0.754 * H / (k_ms * 1e-3) / 23.7e9},
            "valu_issue": valu_issue(k_ms),
            "timing": {"prepare_inputs_s": t_prep, "load_and_upload_s": t_load},
            "check": {"detected_energy_units_last_projection": detected},
        }
        if world == 1 and not args.no_end_to_end:
            out["end_to_end"] = end_to_end_scan(ctx, H, workdir)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx)
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
