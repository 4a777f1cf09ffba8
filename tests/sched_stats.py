"""Diagnostic (GPU): scheduler statistics of the FAST kernel on the bench workload for a set of thresholds."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases
eng = cases.pkg.engine
inp = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mcgpu_bench_512_894/input.in"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
with eng.create(inp, device=0) as ctx:
    img, secs, done = ctx.run_projection(0, n, mode="stats", seed=42)
    s = ctx.scheduler_stats()
    it = s["iterations"]
    print(json.dumps({"thresholds": [os.environ.get(k) for k in ("MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW")],
                      "histories": done, "wave_iterations_per_history": it / done, "mean_flying_lanes": s["flying_lanes"] / it,
                      "compton_lanes_per_round": s["compton_lanes"] / max(s["compton_rounds"], 1), "compton_rounds_per_history": s["compton_rounds"] / done,
                      "rayleigh_lanes_per_round": s["rayleigh_lanes"] / max(s["rayleigh_rounds"], 1), "rayleigh_rounds_per_history": s["rayleigh_rounds"] / done,
                      "new_lanes_per_round": s["new_lanes"] / max(s["new_rounds"], 1), "new_rounds_per_history": s["new_rounds"] / done,
                      "bricks_mixed": ctx.geti("bricks_mixed"), "brick_count": ctx.geti("brick_count"), "blocks_per_cu": ctx.geti("blocks_per_cu")}))
