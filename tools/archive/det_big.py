import sys, hashlib, numpy as np
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
with eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0) as ctx:
    hs = []
    for k in range(4):
        img, secs, done = ctx.run_projection(200, 60_000_000, mode="fast", seed=42)
        hs.append(hashlib.sha256(img.tobytes()).hexdigest()[:16])
    print("bench p200 x4:", hs, "identical" if len(set(hs)) == 1 else "DIFFERENT")
