"""Diagnostic (GPU): where the FAST kernel's wave-time goes on the bench workloads (the MC_STATS build's section counters),
the counterpart of tools/compat_stats.py.  usage: fast_stats.py [workload ...] [--histories N] [--projection P]
Prints one JSON object per workload: share of wave-cycles per section, lanes per batch, events per history."""
import argparse, json, os, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
os.environ.setdefault("MCGPU_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "4d-cbct-mc_amd", "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import cases  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["catphan", "cirs", "thorax"])
ap.add_argument("--histories", type=float, default=5e7)
ap.add_argument("--projection", type=int, default=0)
args = ap.parse_args()
eng = cases.pkg.engine
n = int(args.histories)
for wl in args.workloads:
    wd = Path(os.path.join(tempfile.gettempdir(), f"mcgpu_bench_{wl}_512_894"))
    if not (wd / "input.in").exists():
        wd.mkdir(parents=True, exist_ok=True)
        bench.build_workload(wd, wl, 100_000_000, 894, eng)
    with eng.create(str(wd / "input.in"), device=0) as ctx:
        p = args.projection
        ctx.run_projection(p, n, mode="fast", seed=42)
        _, secs, done = ctx.run_projection(p, n, mode="fast", seed=42)
        ctx.run_projection(p, n, mode="stats", seed=42)
        s = ctx.scheduler_stats()
        it, sp = max(s["iterations"], 1), max(s["scheduling_points"], 1)
        fl, sc = s["cycles_flight"], s["cycles_sched_point"]
        tot = fl + sc
        services = s["cycles_compton"] + s["cycles_rayleigh"] + s["cycles_new"]
        per = lambda k: round(s[k] / done, 4)
        print(json.dumps({
            "workload": wl, "projection": p, "histories": done, "fast_Ghist_per_s": round(done / secs / 1e9, 3),
            "share_of_wave_cycles": {
                "flight": round(fl / tot, 3), "flight_until_voxel_arrives": round(s["cycles_flight_to_voxel"] / tot, 3),
                "flight_resolve": round(s["cycles_flight_resolve"] / tot, 3),
                "settle_hop_and_real": round(s["cycles_settle"] / tot, 3), "compton": round(s["cycles_compton"] / tot, 3),
                "rayleigh": round(s["cycles_rayleigh"] / tot, 3), "tally_source": round(s["cycles_new"] / tot, 3),
                "ballots_trades_exchanges": round((sc - s["cycles_settle"] - services) / tot, 3)},
            "wave_cycles_per_history": round(tot / done, 1),
            "flight": {"iterations_per_history": per("iterations"), "steps_per_history": per("lanes_taking_a_step"),
                       "lanes_still_flying_after_an_iteration": round(s["flying_lanes"] / it, 2), "lanes_taking_a_step_per_iteration": round(s["lanes_taking_a_step"] / it, 2), "iterations_per_sched_point": round(it / sp, 2),
                       "voxel_loads_per_history": per("voxel_load_lanes"), "iterations_with_voxel_load": round(s["iter_with_voxel_load"] / it, 3),
                       "exact_sigma_loads_per_history": per("sigma_load_lanes"), "iterations_with_sigma_load": round(s["iter_with_sigma_load"] / it, 3)},
            "compton": {"batches_per_history": per("compton_rounds"), "lanes_per_batch": round(s["compton_lanes"] / max(s["compton_rounds"], 1), 1),
                        "angle_lanes": round(s["compton_angle_lanes"] / max(s["compton_rounds"], 1), 1),
                        "shell_retry_lanes": round(s["compton_shell_lanes"] / max(s["compton_rounds"], 1), 1),
                        "completed_per_batch": round(s["compton_done_lanes"] / max(s["compton_rounds"], 1), 1),
                        "events_per_history": per("compton_done_lanes"),
                        "trials_per_event": round((s["compton_angle_lanes"] + s["compton_shell_lanes"]) / max(s["compton_done_lanes"], 1), 3)},
            "rayleigh": {"batches_per_history": per("rayleigh_rounds"), "lanes_per_batch": round(s["rayleigh_lanes"] / max(s["rayleigh_rounds"], 1), 1)},
            "tally_source": {"batches_per_history": per("new_rounds"), "lanes_per_batch": round(s["new_lanes"] / max(s["new_rounds"], 1), 1)},
            "sched_points_per_history": per("scheduling_points"), "drain_fraction": round(s["drain_points"] / sp, 3),
            "pool_after_sched_point": {k: round(s[k] / sp, 1) for k in ("pool_flyable", "pool_wants_new", "pool_compton")},
            "slots_traded_per_sched_point": round(s["slots_traded"] / sp, 2)}), flush=True)
