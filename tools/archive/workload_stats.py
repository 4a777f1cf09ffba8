#!/usr/bin/env python3
"""Dev measurement (GPU): FAST kernel throughput and scheduler statistics (diagnostic build) per bench workload.
Usage: python tools/workload_stats.py [catphan cirs thorax]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
__import__("os").environ.setdefault("MCGPU_AMD_LIB", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "4d-cbct-mc_amd", "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import bench
import cases

eng = cases.pkg.engine
for wl in (sys.argv[1:] or ["catphan", "cirs", "thorax"]):
    wd = Path(f"/tmp/mcgpu_wl_{wl}"); wd.mkdir(exist_ok=True)
    t0 = time.time()
    inp = wd / "input.in"
    if not inp.exists():
        inp = bench.build_workload(wd, wl, int(1e8), 894, eng)
    t1 = time.time()
    with eng.create(inp, device=0) as ctx:
        out = {"workload": wl, "prepare_s": round(t1 - t0, 1), "load_s": round(time.time() - t1, 2), "materials": ctx.geti("num_materials_used"),
               "bricks": ctx.geti("brick_count"), "brick_shift": ctx.geti("brick_shift"), "mixed": ctx.geti("bricks_mixed"), "exterior": ctx.geti("bricks_exterior")}
        for p in (0, 223, 447):
            ctx.run_projection(p, int(2e7), mode="fast", seed=42)
            _, secs, done = ctx.run_projection(p, int(1e8), mode="fast", seed=42)
            out[f"p{p}_Ghist_per_s"] = round(done / secs / 1e9, 3)
        out["lds"] = ctx.geti("lds_bytes_fast"); out["wg_per_cu"] = ctx.geti("blocks_per_cu"); out["sig_shift"] = ctx.geti("sigma_bracket_shift")
        _, secs, done = ctx.run_projection(0, int(3e7), mode="stats", seed=42)
        s = ctx.scheduler_stats()
        out["iter_per_hist"] = round(s["iterations"] / done, 4)
        out["flying"] = round(s["flying_lanes"] / max(s["iterations"], 1), 1)
        out["sched_points_per_hist"] = round(s["scheduling_points"] / done, 4)
        for k in ("compton", "rayleigh", "new"):
            out[f"{k}_rounds_per_hist"] = round(s[f"{k}_rounds"] / done, 4)
            out[f"{k}_lanes"] = round(s[f"{k}_lanes"] / max(s[f"{k}_rounds"], 1), 1)
        out["compton_angle_shell_done_lanes"] = [round(s[k] / max(s["compton_rounds"], 1), 1) for k in ("compton_angle_lanes", "compton_shell_lanes", "compton_done_lanes")]
        out["cycles_per_hist"] = {k[7:]: round(s[k] / done) for k in s if k.startswith("cycles_")}
        sp = max(s["scheduling_points"], 1)
        out["after_sched_point"] = {k: round(s[k] / sp, 1) for k in ("pool_flyable", "pool_wants_new", "pool_compton", "slots_traded", "take_lanes")}
        out["voxel_load_iter_frac"] = round(s["iter_with_voxel_load"] / max(s["iterations"], 1), 3)
        out["sigma_load_iter_frac"] = round(s["iter_with_sigma_load"] / max(s["iterations"], 1), 3)
    print(json.dumps(out))
