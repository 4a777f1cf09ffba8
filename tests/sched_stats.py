"""Diagnostic (GPU): scheduler statistics of the FAST kernel on the bench workload for a set of tuning parameters.
usage: sched_stats.py [input.in] [histories] ["tC,tR,tN,flyable_low,swap_batch" ...]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCGPU_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "4d-cbct-mc_amd", "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import cases
eng = cases.pkg.engine
inp = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mcgpu_bench_catphan_512_894/input.in"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
configs = sys.argv[3:] or [""]
KEYS = ("MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW", "MCGPU_FLYABLE_LOW", "MCGPU_SWAP_BATCH")
with eng.create(inp, device=0) as ctx:
    for cfg in configs:
        for k in KEYS:
            os.environ.pop(k, None)
        for k, v in zip(KEYS, [x for x in cfg.split(",") if x]):
            os.environ[k] = v
        ctx.reload_env_knobs()
        ctx.run_projection(0, n, mode="fast", seed=42)  # warm
        _, secs, _ = ctx.run_projection(0, n, mode="fast", seed=42)
        img, secs_stats, done = ctx.run_projection(0, n, mode="stats", seed=42)
        s = ctx.scheduler_stats()
        it = max(s["iterations"], 1)
        print(json.dumps({"cfg": cfg, "Mhist_per_s": round(done / secs / 1e6, 1), "iter_per_hist": round(it / done, 4),
                          "mean_flying": round(s["flying_lanes"] / it, 2),
                          "compton": [round(s["compton_lanes"] / max(s["compton_rounds"], 1), 1), round(s["compton_rounds"] / done, 5)],
                          "rayleigh": [round(s["rayleigh_lanes"] / max(s["rayleigh_rounds"], 1), 1), round(s["rayleigh_rounds"] / done, 5)],
                          "new": [round(s["new_lanes"] / max(s["new_rounds"], 1), 1), round(s["new_rounds"] / done, 5)],
                          "sched_points_per_hist": round(s["scheduling_points"] / done, 4),
                          "take": [round(s["take_lanes"] / max(s["take_rounds"], 1), 1), round(s["take_rounds"] / done, 5)],
                          "drain_frac": round(s["drain_points"] / max(s["scheduling_points"], 1), 3),
                          "compton_trials": {"angle_lanes_per_round": round(s["compton_angle_lanes"] / max(s["compton_rounds"], 1), 1),
                                             "shell_lanes_per_round": round(s["compton_shell_lanes"] / max(s["compton_rounds"], 1), 1),
                                             "done_per_round": round(s["compton_done_lanes"] / max(s["compton_rounds"], 1), 1)},
                          "pool_after_sched_point": {k: round(s[k] / max(s["scheduling_points"], 1), 1) for k in
                                                     ("pool_flyable", "pool_wants_new", "pool_compton")},
                          "per_history": {"flight_steps": round(s["flying_lanes"] / done, 3), "voxel_loads": round(s["voxel_load_lanes"] / done, 3),
                                          "exact_sigma_loads": round(s["sigma_load_lanes"] / done, 3),
                                          "real_records": round((s["compton_done_lanes"]) / done, 3)},
                          "flight_loads": {"iter_with_voxel_load": round(s["iter_with_voxel_load"] / it, 3), "voxel_lanes_per_iter": round(s["voxel_load_lanes"] / it, 2),
                                           "iter_with_sigma_load": round(s["iter_with_sigma_load"] / it, 3), "sigma_lanes_per_iter": round(s["sigma_load_lanes"] / it, 2)},
                          "cycles_per_hist": {k[7:]: round(s[k] / done, 1) for k in ("cycles_flight", "cycles_compton", "cycles_rayleigh", "cycles_new")},
                          "wave_cycles_per_hist_total(100MHz ticks?)": round(secs_stats * 1e8 * 6144 / done, 1),
                          "blocks_per_cu": ctx.geti("blocks_per_cu"), "lds": ctx.geti("lds_bytes_fast")}))
