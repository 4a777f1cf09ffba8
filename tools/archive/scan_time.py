import sys, json, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
out = Path("/tmp/scan_t"); out.mkdir(exist_ok=True)
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    for kw in (dict(write_stacks=True), dict(write_stacks=False), dict(write_stacks=True, write_ascii=True)):
        r = ctx.run_scan(mode="fast", first_projection=100, num_projections=8, histories=int(1e8), crop_nx=1024, output_folder=out, **kw)
        print(kw, {k: round(v, 4) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"})
