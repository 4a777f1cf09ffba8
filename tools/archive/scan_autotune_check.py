import sys, time, os
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
inp = sys.argv[1]
out = Path("/tmp/scan_at"); out.mkdir(exist_ok=True)
for tune in (True, False):
    if not tune: os.environ["MCGPU_NO_AUTOTUNE"] = "1"
    with eng.create(inp, device=0) as ctx:
        r = ctx.run_scan(mode="fast", first_projection=0, num_projections=60, histories=int(1e8), crop_nx=1024, write_stacks=False, output_folder=out)
        print("autotune" if tune else "default ", round(r["seconds_kernels"] / 60 * 1e3, 3), "ms per projection")
