"""Dev check (GPU): class sums of the two FAST schedulers on a small case, with and without the exterior hop."""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
eng = cases.pkg.engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
with tempfile.TemporaryDirectory() as tmp:
    d = cases.build_case("catphan64", Path(tmp) / "c")
    for noext in (0, 1):
        for sched in (0, 1):
            os.environ["MCGPU_FAST_SCHED"] = str(sched)
            if noext: os.environ["MCGPU_NO_EXTERIOR"] = "1"
            else: os.environ.pop("MCGPU_NO_EXTERIOR", None)
            with eng.create(d, device=0) as ctx:
                img, secs, done = ctx.run_projection(0, n, mode="fast", seed=5)
                print("noext", noext, "sched", sched, "done", done, "sums", [int(img[k].sum()) for k in range(4)], "word0", int(img.reshape(-1)[0]), "s", round(secs, 4), flush=True)
