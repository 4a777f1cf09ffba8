import sys, json, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
out = Path("/tmp/scan_full"); out.mkdir(exist_ok=True)
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    t0 = time.time()
    r = ctx.run_scan(mode="fast", histories=int(1e8), crop_nx=1024, write_stacks=False, output_folder=out)
    print("894 projections, no output:", {k: round(v, 3) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"}, "wall", round(time.time() - t0, 2))
    t0 = time.time()
    r = ctx.run_scan(mode="fast", first_projection=0, num_projections=120, histories=int(1e8), crop_nx=1024, write_stacks=True, output_folder=out, pixel_spacing=(0.776, 0.776))
    print("120 projections with stacks:", {k: round(v, 3) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"}, "wall", round(time.time() - t0, 2))
    s = eng.stack_read(out / "projections_total.mha")
    print("stack", s.shape, float(s.min()), float(s.max()), float(s.mean()))
