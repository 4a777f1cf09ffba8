"""One COMPAT projection of a bench workload (for rocprofv3 --pmc): compat_one.py <workload dir> [histories]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
with eng.create(sys.argv[1] + "/input.in", device=0) as ctx:
    batches, hpt, total = ctx.reference_shape(int(float(sys.argv[2]) if len(sys.argv) > 2 else 2e7))
    img, secs, done = ctx.run_projection(300, batches, mode="compat", seed=42, hpt=hpt)
    print(f"{done} histories in {secs * 1e3:.1f} ms")
