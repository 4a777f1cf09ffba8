"""Dev check (GPU): the whole 894-projection scan of the bench workload in COMPAT mode (RANECU streams, the reference's arithmetic,
seed stepped per projection like update_seed_PRNG), stacks on disk.  Usage: python tools/full_scan_compat.py [workload]"""
import sys, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
from pathlib import Path
eng = cases.pkg.engine
wl = sys.argv[1] if len(sys.argv) > 1 else "catphan"
out = Path(f"/tmp/scan_full_compat_{wl}"); out.mkdir(exist_ok=True)
with eng.create(f"/tmp/mcgpu_bench_{wl}_512_894/input.in", device=0) as ctx:
    t0 = time.time()
    r = ctx.run_scan(mode="compat", histories=int(1e8), crop_nx=1024, write_stacks=True, output_folder=out, pixel_spacing=(0.776, 0.776))
    print(wl, "894 projections, COMPAT, stacks on disk:", {k: round(v, 3) if isinstance(v, float) else v for k, v in r.items() if k != "zero_replacement"}, "wall", round(time.time() - t0, 2))
    s = eng.stack_read(out / "projections_total.mha")
    print("stack", s.shape, float(s.min()), float(s.max()), float(s.mean()))
    for f in out.glob("*.mha"):
        f.unlink()
