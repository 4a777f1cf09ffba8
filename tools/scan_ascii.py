"""Scan with the reference's ASCII projection files written (63 MB of text per projection): writer time per projection."""
import sys, time, shutil
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
out = Path("/tmp/scan_ascii"); out.mkdir(exist_ok=True)
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    t0 = time.time()
    r = ctx.run_scan(mode="fast", first_projection=0, num_projections=n, histories=int(1e8), crop_nx=1024, write_ascii=True, write_stacks=False, output_folder=out)
    print({k: round(v, 3) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"},
          "per projection: total", round(r["seconds_total"] / n * 1e3, 2), "ms, writer", round(r["seconds_writer"] / n * 1e3, 2), "ms")
shutil.rmtree(out, ignore_errors=True)
