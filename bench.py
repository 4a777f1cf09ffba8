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
the exposed add), `check.sharded_equals_single` (the summed sharded tally against one rank simulating the same history
ids alone, bit for bit) and `collectives` (a short leg per route after the timed region: the exchange and north_star's literal
per-projection `ncclReduce`, the peer-access matrix and each rank's device -- one `--gpus N` command yields both).

This file holds the headline path and the JSON line; the legs live in `bench_legs/` (common: workloads and kernel identity,
roofline, cpu, checks, legs, multi: the N > 1 tally routes, launch: starting the ranks).
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

# the names tests/ and tools/ import from this module
from bench_legs.checks import entry_face_deficit, fast_vs_compat_check, oracle_check  # noqa: E402,F401
from bench_legs.common import (HBM_PEAK_GBS, KERNEL_SOURCES, WORKLOADS, build_workload, kernel_source_hash, kernel_variant,  # noqa: E402,F401
                               knob_environment, package, usable_cpus)
from bench_legs.cpu import cpu_baseline  # noqa: E402,F401
from bench_legs.launch import spawn_ranks  # noqa: E402
from bench_legs.legs import cirs_4d_leg, compat_leg, end_to_end_scan, fdk_leg, other_workloads, text_geometry_load  # noqa: E402,F401
from bench_legs.roofline import measured_ceilings, pmc_summary, roofline_block, timed_launches  # noqa: E402,F401


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--histories", type=float, default=1e8, help="histories per projection per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="catphan")
    ap.add_argument("--voxels", type=int, default=512, help="catphan workload: cube edge")
    ap.add_argument("--projections", type=int, default=894)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="host time budget of the cpu_baseline leg (bounded sample of the same workload)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the pipelined-scan measurements after the timed region")
    ap.add_argument("--no-compat", action="store_true", help="skip the COMPAT-personality leg")
    ap.add_argument("--no-reference-arithmetic", action="store_true", help="skip the fast64 launches (the same kernel with the reference's double-precision sub-steps)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the CIRS / thorax / textured-thorax legs (configs 3-5)")
    ap.add_argument("--no-fdk", action="store_true", help="skip the FDK reconstruction leg (config 4)")
    ap.add_argument("--no-collectives", action="store_true", help="N > 1: skip the exchange-vs-RCCL comparison leg after the timed region")
    ap.add_argument("--scan-projections", type=int, default=894, help="projections of the end_to_end leg")
    ap.add_argument("--ascii-projections", type=int, default=894, help="projections of the end_to_end_ascii leg (63 MB of text each; 0: skip)")
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args()
    if args.gpus is None:  # `torchrun ... bench.py` without --gpus: the launcher's rank count is the GPU count
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    return args


def init_ranks(args, torch):
    """(rank, world, device, dist, backend, ctl, shares_gpus): one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the launcher."""
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
    # route after an RCCL collective has failed (the verdicts are a few bytes of host data)
    ctl = dist.new_group(backend="gloo") if (dist and backend == "nccl") else None
    return rank, world, device, dist, backend, ctl, bool(share and world > n_dev)


def main():
    args = parse_args()
    # No launcher gave this process a rank: it becomes the launcher (nothing GPU-related has been imported yet).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner from C stdio on
    # its first collective), so everything else is sent to stderr and the line goes to the original descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank, world, device, dist, backend, ctl, shares_gpus = init_ranks(args, torch)

    from bench_legs.multi import TallyRoute
    pkg = package()
    eng = pkg.engine
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
    ctx = eng.create(inp, device=device)
    t_load = time.perf_counter() - t_load0
    nz, nx = ctx.detector_shape
    nproj = ctx.num_projections
    stream = torch.cuda.current_stream().cuda_stream

    # ---- the timed region: W untimed steps, then exactly K steps between barrier + synchronize, max over ranks
    route = TallyRoute(pkg=pkg, torch=torch, dist=dist, ctl=ctl, backend=backend, rank=rank, world=world, device=device, ctx=ctx,
                       stream=stream, args=args, H=H)
    for i in range(args.warmup):
        route.step(i, False)
    route.drain()
    route.barrier()
    torch.cuda.synchronize()
    route.reduce_bytes = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        route.step(args.warmup + i, True)
    route.drain()
    route.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = route.kernel_ms()
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    multi = route.check(float(np.mean(kernel_ms))) if dist else None
    collectives = route.collectives_comparison() if (dist and not args.no_collectives) else None
    detected = 0
    if route.last_reduced is not None and (not dist):
        detected = int(route.last_reduced.sum().item())

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
            # what "f32" narrows against the reference, and the same kernel WITHOUT that narrowing (value_reference_arithmetic below)
            "dtype_note": "the reference computes three sub-steps in double -- rotate_double (MC-GPU_kernel_v1.3.cu:1103-1148), GRAa (:1181-1246), GCOa's cdt1 / costh "
                          "chain (:1329-1331, :1372, :1427) -- which `value` (mode fast) computes in float32, validated statistically against the bit-exact personality; "
                          "everything else is float32 in the reference too (its GPU build: -use_fast_math).  value_reference_arithmetic = the same kernel, scheduler and "
                          "random-number streams with those three sub-steps in double as the reference has them (mode fast64, csrc/track_fast64.hip); compat.value = the "
                          "reference's arithmetic AND its RANECU streams, bit-identical to the CPU oracle",
            "config": {"workload": f"{label}_{args.projections}proj_{H:.0e}hist_per_proj_per_gpu",
                       "detector": f"{nx}x{nz}", "histories_per_projection_per_gpu": H, "kernel": "fast",
                       "parallelism": route.parallelism_text(),
                       "parallelism_route": route.kind, "parallelism_fallbacks": route.fallbacks if dist else None,
                       "ranks_started_by": "bench.py itself" if os.environ.get("BENCH_SPAWNED") else ("an external launcher" if dist else "single process"),
                       "process_group_backend": backend, "ranks_share_gpus": shares_gpus,
                       "volume_kind": ["u8-palette", "u16-palette", "raw-float2"][ctx.geti("volume_kind")],
                       "volume_bytes": ctx.geti("volume_bytes_device"), "materials_used": ctx.geti("num_materials_used"),
                       "lds_bytes_per_workgroup": ctx.geti("lds_bytes_fast"), "workgroups_per_cu": ctx.geti("blocks_per_cu"),
                       "second_level": "tile records" if ctx.geti("tile_records") else ("4-bit codes" if ctx.geti("sub_brick_table") else "none"), "tiles_in_mixed_bricks": ctx.geti("tiles_in_mixed_bricks"),
                       "fast_scheduler": "workgroup-level pool" if kernel_variant(args.workload, ctx)["fast_scheduler"] else "per-wave pools",
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
            out["check"]["projection_sharded_equals_single" if route.kind == "none" else "sharded_equals_single"] = v
            out["check"]["passed"] = bool(v and v["passed"])
            failed = failed or not out["check"]["passed"]
            out["reduce"] = route.report(multi, elapsed, args.steps)
            if collectives:
                collectives["kernel_ms_avg_of_the_timed_region"] = k_ms
                collectives["exposed_ms_per_step"] = {k: (None if v_ is None else v_ - k_ms) for k, v_ in collectives["ms_per_step"].items()}
                out["reduce"]["routes"] = collectives
                route_ok = all(v is None or v["passed"] for v in collectives.get("sharded_equals_single", {}).values())
                out["check"]["every_route_sharded_equals_single"] = route_ok
                out["check"]["passed"] = bool(out["check"]["passed"] and route_ok)
                failed = failed or not route_ok
        if world == 1:
            # the headline kernel at the reference's arithmetic: same launches (warm-up + steps at the same angles), same HIP-event timing
            if not args.no_reference_arithmetic:
                k64, k64_min, _ = timed_launches(ctx, torch, H, launches=args.steps, warm=args.warmup, mode="fast64")
                out["value_reference_arithmetic"] = H / (k64 * 1e-3)
                out["reference_arithmetic"] = {"mode": "fast64", "kernel": "track_pool_kernel<..., 1> (csrc/track_fast64.hip)", "kernel_ms_avg": k64, "kernel_ms_min": k64_min,
                                               "launches": args.steps, "cost_over_value": k64 / k_ms - 1.0, "unit": "histories/s"}
            if not args.no_compat:
                out["compat"] = compat_leg(ctx, torch, H)
                both = fast_vs_compat_check(ctx, modes=("fast",) if args.no_reference_arithmetic else ("fast", "fast64"))
                out["check"]["fast_vs_compat"] = both if args.no_reference_arithmetic else both["fast"]
                failed = failed or not out["check"]["fast_vs_compat"]["passed"]
                if not args.no_reference_arithmetic:  # the reference-arithmetic variant against the bit-exact personality, same COMPAT launches
                    out["check"]["fast64_vs_compat"] = both["fast64"]
                    failed = failed or not both["fast64"]["passed"]
            if not args.no_end_to_end:
                out["end_to_end"] = end_to_end_scan(ctx, H, workdir, n=min(args.scan_projections, nproj))
                # the 894-projection scan with stacks, sustained: what the headline's launches become over a whole trajectory
                out["sustained_value"] = out["end_to_end"]["histories_per_s_with_stacks"]
                if args.ascii_projections > 0:
                    out["end_to_end_ascii"] = end_to_end_scan(ctx, H, workdir, n=min(args.ascii_projections, nproj), ascii_files=True)
                # the same context creation from the reference's own text geometry (sidecar ignored)
                out["timing"]["load_and_upload_s_text_geometry"] = text_geometry_load(eng, inp, device)
                out["timing"]["geometry_text_bytes"] = (workdir / "geometry.vox").stat().st_size
            if not args.no_workloads and args.workload == "catphan":
                out["workloads"] = other_workloads(eng, torch, H, args.projections, device, ceilings, scans=not args.no_end_to_end)
                kd = out["workloads"]["thorax"].get("entry_face_shell")
                out["check"]["known_deviations"] = {"beam_edge_column": (out["check"].get("fast_vs_compat") or {}).get("beam_edge_column"), "entry_face": kd}
                failed = failed or not (kd is None or kd["passed"])
                if not args.no_fdk:
                    out["fdk"] = fdk_leg(pkg, device)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only (the other ranks would idle meanwhile)
            base, img_cpu, w2_cpu, n_cpu = cpu_baseline(ctx, seconds_budget=args.cpu_seconds)
            out["cpu_baseline"] = base
            # a second ceiling: one scattered 64-bit atomic add per detected photon, rate measured in this run
            tally_hits = base["events_per_history"]["tally_hits"]
            a_peak = ceilings["scattered_64bit_atomic_adds_per_s"] / 1e9 if ceilings else 23.7  # 23.7: tools/archive/micro/atomic_rate.hip, round 2
            out["atomic_roofline"] = {"bound": "scattered 64-bit atomic adds", "achieved": tally_hits * H / (k_ms * 1e-3) / 1e9, "peak": a_peak,
                                      "peak_source": "measured in this run (mcgpu_microbench)" if ceilings else "tools/archive/micro/atomic_rate.hip, round 2",
                                      "unit": "Gatomic/s", "frac": tally_hits * H / (k_ms * 1e-3) / (a_peak * 1e9),
                                      "detected_photons_per_history": tally_hits}
            out["check"].update(oracle_check(ctx, H, img_cpu, w2_cpu, n_cpu))
            out["check"]["passed"] = bool(out["check"]["passed"] and out["check"].get("fast_vs_compat", {}).get("passed", True) and
                                          out["check"].get("fast64_vs_compat", {}).get("passed", True))
            failed = failed or not out["check"]["passed"]
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    route.barrier()
    route.close()
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: the correctness check FAILED (see `check` in the line above)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
