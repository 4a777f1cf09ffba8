#!/usr/bin/env python3
"""Dev check (GPU): FAST vs oracle per scatter class on a bench workload with a large oracle sample.
Usage: python tools/bias_check.py thorax 223 [oracle_batches] [gpu_histories]"""
import sys, os, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, cases, oracle_lib as ol, parity
wl, p = sys.argv[1], int(sys.argv[2])
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000
ng = int(float(sys.argv[4])) if len(sys.argv) > 4 else 400_000_000
eng = cases.pkg.engine
wd = Path(f"/tmp/mcgpu_wl_{wl}"); wd.mkdir(exist_ok=True)
inp = wd / "input.in"
if not inp.exists():
    inp = bench.build_workload(wd, wl, int(1e8), 894, eng)
with eng.create(inp, device=0) as ctx:
    T = parity.tables_from_context(ctx)
    t0 = time.time()
    img_cpu, w2, _ = T.track_with_variance(p, int(os.environ.get("ORACLE_SEED", "777")), 0, nb, 150, ol.MATH_LIBM, n_threads=min(16, len(os.sched_getaffinity(0))))
    print("oracle", nb * 150, "histories in", round(time.time() - t0, 1), "s")
    img = np.zeros((4,) + ctx.detector_shape, dtype=np.uint64); done = 0
    for k in range(4):
        part, _, d = ctx.run_projection(p, ng // 4, mode="fast", seed=int(os.environ.get("GPU_SEED", "300")) + k); img += part; done += d
    img_cpu, w2 = img_cpu.reshape(img.shape), w2.reshape(img.shape)
    zs = parity.class_energy_z(img, done, img_cpu, w2, nb * 150)
    print("ratios", [round(float(img[k].sum() / done / (img_cpu[k].sum() / (nb * 150))), 5) for k in range(4)], "z", np.round(zs, 2).tolist())
    z, m = parity.measured_z(parity.blocks(img, 16), done, parity.blocks(img_cpu, 16), parity.blocks(w2, 16), nb * 150)
    for k in range(4):
        zk = z[k][m[k]]
        print("class", k, "blocks", zk.size, "mean z", round(float(zk.mean()), 4) if zk.size else None, "std", round(float(zk.std()), 4) if zk.size else None)
