"""Event-queue kernel vs lane-bound kernel (GPU): identical tallies on small cases, then timing on the bench workload."""
import os, sys, hashlib, time, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, cases
eng = cases.pkg.engine
base = Path(tempfile.mkdtemp(prefix="qchk_"))
ok = True
stage = sys.argv[1] if len(sys.argv) > 1 else "all"
small = [("water", 50_000), ("catphan64", 400_000), ("catphan64_ct", 200_000), ("slab_angles", 200_000), ("graded_u16", 200_000)]
for name, n in small:
    inp = cases.build_case(name, base / name)
    with eng.create(inp, device=0) as ctx:
        for p in range(min(ctx.num_projections, 2)):
            os.environ.pop("MCGPU_FAST_KERNEL", None)
            a, _, da = ctx.run_projection(p, n, mode="fast", seed=42 + p)
            os.environ["MCGPU_FAST_KERNEL"] = os.environ.get("QCHK_KERNEL", "queue")
            b, _, db = ctx.run_projection(p, n, mode="fast", seed=42 + p)
            same = np.array_equal(a, b)
            ok &= same
            print(name, p, "same" if same else f"DIFFERENT sum {int(a.sum())} vs {int(b.sum())} done {da} {db}", flush=True)
    if stage == "tiny":
        break
print("ALL SAME" if ok else "MISMATCH", flush=True)
