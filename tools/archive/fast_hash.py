#!/usr/bin/env python3
"""Dev regression (GPU): SHA-256 of FAST-mode tallies on the small cases + the bench geometry.  A scheduling-only
change of the kernel must leave every hash unchanged (per-history RNG streams, integer tallies)."""
import hashlib, json, os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
eng = cases.pkg.engine
out = {}
base = Path(tempfile.mkdtemp(prefix="fasthash_"))
for name, n in [("air", 100_000), ("water", 300_000), ("catphan64", 400_000), ("catphan64_ct", 200_000), ("slab_angles", 200_000)]:
    inp = cases.build_case(name, base / name)
    with eng.create(inp, device=0) as ctx:
        for p in range(ctx.num_projections):
            img, secs, done = ctx.run_projection(p, n, mode="fast", seed=42 + p)
            out[f"{name}:{p}"] = hashlib.sha256(img.tobytes()).hexdigest()[:16] + f":{int(img.sum())}"
bench = Path("/tmp/mcgpu_bench_catphan_512_894/input.in")
if bench.exists():
    with eng.create(bench, device=0) as ctx:
        for p in (0, 300):
            img, secs, done = ctx.run_projection(p, 3_000_000, mode="fast", seed=42)
            out[f"bench:{p}"] = hashlib.sha256(img.tobytes()).hexdigest()[:16] + f":{int(img.sum())}"
print(json.dumps(out, indent=1))
