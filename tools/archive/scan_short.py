"""Short scan (no output files) for timeline profiling of the scan driver: python tools/scan_short.py [n_projections]."""
import sys, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out = Path("/tmp/scan_short"); out.mkdir(exist_ok=True)
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    r = ctx.run_scan(mode="fast", first_projection=0, num_projections=n, histories=int(1e8), crop_nx=1024, write_stacks=False, output_folder=out)
    print({k: round(v, 4) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"})
