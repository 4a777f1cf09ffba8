import sys
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    for mode in ("compat", "fast"):
        ctx.run_projection(0, int(2e7), mode=mode, seed=42)
        _, secs, done = ctx.run_projection(0, int(1e8), mode=mode, seed=42)
        print(mode, done, "histories in", round(secs * 1e3, 2), "ms ->", round(done / secs / 1e9, 3), "e9 hist/s")
