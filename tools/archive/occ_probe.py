import sys
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    ctx.run_projection(0, int(2e7), mode="fast", seed=1)
    _, s, d = ctx.run_projection(0, int(1e8), mode="fast", seed=1)
    print("wg_per_cu", ctx.geti("blocks_per_cu"), "lds", ctx.geti("lds_bytes_fast"), "Ghist/s", d / s / 1e9)
