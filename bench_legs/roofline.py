"""`roofline` / `valu_issue` objects of the JSON line: algorithmic bytes over the measured kernel time, the stamped PMC summaries, the in-run ceilings."""
from __future__ import annotations

import json

import numpy as np

from .common import HBM_PEAK_GBS, ROOT, WORKLOADS, kernel_source_hash, kernel_variant, knob_environment

def pmc_summary(workload: str, ctx=None):
    """The committed rocprofv3 PMC summary of the FAST kernel (separate --pmc passes, tools/pmc_collect.sh), accepted only if
    it was collected from THIS kernel build (source hash) on THIS workload, for the kernel variant this context dispatches
    and under the same MCGPU_* knobs; else (None, reason)."""
    f = ROOT / "profiles" / ("pmc_summary_latest.json" if workload == "catphan" else f"pmc_summary_{workload}.json")
    if not f.exists():
        return None, "no summary committed"
    d = json.loads(f.read_text())
    stamp = d.get("_stamp", {})
    if stamp.get("kernel_source_sha16") != kernel_source_hash():
        return None, f"stale: summary is of kernel build {stamp.get('kernel_source_sha16')}, running {kernel_source_hash()}"
    if stamp.get("workload") != workload:
        return None, f"stale: summary is of workload {stamp.get('workload')}"
    if ctx is not None and stamp.get("variant") != kernel_variant(workload, ctx):
        return None, f"stale: summary is of kernel variant {stamp.get('variant')}, this context dispatches {kernel_variant(workload, ctx)}"
    if stamp.get("knobs", {}) != knob_environment():
        return None, f"stale: summary was collected under knobs {stamp.get('knobs')}, running under {knob_environment()}"
    return d, f"{f.name} ({stamp.get('collected', '?')})"

def timed_launches(ctx, torch, H, launches=8, warm=2, mode="fast"):
    """Mean kernel time [ms] of `launches` FAST (`mode`: "fast" / "fast64") launches of H histories (HIP events on the launch stream), after `warm` untimed."""
    nz, nx = ctx.detector_shape
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed, nproj, ms = ctx.geti("seed"), ctx.num_projections, []
    for i in range(warm + launches):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch((i * 149) % nproj, image.data_ptr(), H, mode=mode, seed=seed, stream=stream)
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

def kernel_name(workload, ctx=None):
    """The kernel the engine dispatches for this context, as a profiler shows it: track_pool_kernel<volume kind, segment loop, 0> (the
    per-wave pools of track_fast.hip; last argument 1 = track_fast64.hip) or track_wg_kernel<volume kind> (MCGPU_FAST_SCHED=1)."""
    if ctx is None:
        return "track_pool_kernel<...> (fast)"
    v = kernel_variant(workload, ctx)
    vk = 4 if v["tile_records"] else (3 if ctx.geti("sub_brick_table") else v["volume_kind"])
    what = {0: "u8 volume", 1: "u16 volume", 2: "raw float2 volume", 3: "u8 volume + 4-bit sub-brick codes", 4: "u8 volume + tile records"}[vk]
    if v["fast_scheduler"]:
        return f"track_wg_kernel<{vk}> (fast, workgroup-level pool, {what})"
    return f"track_pool_kernel<{vk}, {'true' if v['segment_loop'] else 'false'}, 0> (fast, {what}{', flight segment as an inner loop' if v['segment_loop'] else ''})"


def roofline_block(workload, H, k_ms, ceilings=None, ctx=None):
    """`roofline` object of one workload: algorithmic bytes of the reference layout over the measured kernel time, plus the
    PMC-counter traffic of this kernel build when a stamped summary of it is committed."""
    label, algo_bytes, algo_src = WORKLOADS[workload]
    achieved = algo_bytes * H / (k_ms * 1e-3) / 1e9
    pmc, pmc_src = pmc_summary(workload, ctx) if H == int(1e8) else (None, "summary is per 1e8-history launch")
    traffic = traffic_raw = hbm_counter_frac = valu = l2_hit = None
    if pmc:
        # FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read, so
        # the guide doubles it (MI355X_MICROARCH.md, HBM section) -- a correction it establishes for 16 B-per-lane streams only and
        # calls uncalibrated for other widths.  This kernel's reads are scattered 1-, 8- and 16-byte gathers: `traffic` follows the
        # guide (doubled: an upper bracket if such a gather is a 64-B request), `traffic_uncorrected` is the raw counter (the lower one).
        traffic = (2.0 * pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
        traffic_raw = (pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
        hbm_counter_frac = traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        insts = pmc["SQ_INSTS_VALU"]["mean_per_dispatch"]
        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
            l2_hit = pmc["TCC_HIT_sum"]["mean_per_dispatch"] / max(pmc["TCC_HIT_sum"]["mean_per_dispatch"] + pmc["TCC_MISS_sum"]["mean_per_dispatch"], 1.0)
        # ceiling: a dense dependent-FMA kernel, 8 waves/SIMD, 16-32 active lanes, on the same chip (tools/archive/micro/exec_skip.hip:
        # 5.24e9 wave-instructions on 1024 SIMDs in 5.1 ms)
        # measured in this run when `ceilings` is given (the highest of the three lane masks: the conservative peak); else the
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
            "traffic": traffic, "traffic_uncorrected": traffic_raw, "hbm_counter_frac": hbm_counter_frac, "traffic_source": pmc_src,
            "traffic_note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE as the guide prescribes for gfx950; the factor 2 is calibrated for wide coalesced reads, this kernel gathers 1-16 bytes: the truth lies between traffic_uncorrected and traffic",
            "fabric_bytes_per_history": None if traffic is None else traffic / H, "l2_hit_rate": l2_hit,
            "kernel": kernel_name(workload, ctx),
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
