import sys, numpy as np, tempfile
from pathlib import Path
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
inp = cases.build_case("catphan64_dose", Path(tempfile.mkdtemp()))
with eng.create(inp, device=0) as ctx:
    res = []
    for k in range(6):
        ctx.dose_clear()
        img, _, _ = ctx.run_projection(0, 6_000_000, mode="fast", seed=7)
        v, m = ctx.dose_read()
        res.append((img.copy(), v.copy(), m.copy()))
    for k in range(1, 6):
        print("run", k, "image diff", np.count_nonzero(res[k][0] != res[0][0]), "vox diff", np.count_nonzero(res[k][1] != res[0][1]), "mat diff", np.count_nonzero(res[k][2] != res[0][2]),
              "sums", int(res[k][1][...,0].sum()) - int(res[0][1][...,0].sum()), int(res[k][2][:,0].sum()) - int(res[0][2][:,0].sum()))
    d = np.argwhere(res[1][1] != res[0][1])[:5]
    for i in d: print(tuple(i), res[0][1][tuple(i)], res[1][1][tuple(i)])
    print(res[0][2][:8], res[1][2][:8])
