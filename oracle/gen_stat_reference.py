#!/usr/bin/env python3
"""Statistical reference for the FAST kernel: K independent runs of the CPU oracle in its LIBM mode
(pinned bit-identical to the reference build by tests/test_oracle_golden.py) on one case and projection -> per-pixel mean and
run-to-run variance of the four class images.  usage: gen_stat_reference.py [case [projection [batches_per_run]]]
(defaults: catphan64 0 266667 -> tests/golden/stat_catphan64.npz; other cases -> stat_<case>_p<projection>.npz).
Test infrastructure."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT))
import cases, oracle_lib as ol, parity, tempfile
CASE = sys.argv[1] if len(sys.argv) > 1 else "catphan64"
PROJ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K, NB, HPT = 16, (int(sys.argv[3]) if len(sys.argv) > 3 else 266667), 150  # 16 x 4.0e7 histories by default
OUT = "stat_catphan64.npz" if (CASE, PROJ) == ("catphan64", 0) else f"stat_{CASE}_p{PROJ}.npz"
eng = cases.pkg.engine
with tempfile.TemporaryDirectory() as wd:
    inp = cases.build_case(CASE, wd)
    with eng.create(inp, device=-1) as ctx:
        T = parity.tables_from_context(ctx)
        nz, nx = ctx.detector_shape
        runs = []
        t0 = time.time()
        for k in range(K):
            img, _ = T.track(PROJ, 1000 + 7919 * k, 0, NB, HPT, ol.MATH_LIBM, n_threads=8)
            runs.append(img.reshape(4, nz, nx).astype(np.float64) / (NB * HPT))
            print(f"run {k}: {time.time() - t0:.0f} s", flush=True)
runs = np.stack(runs)
# 3x3 pixel blocks keep the fixture small and the per-block counts high
b = runs[:, :, : nz // 3 * 3, : nx // 3 * 3].reshape(K, 4, nz // 3, 3, nx // 3, 3).sum(axis=(3, 5))
np.savez_compressed(ROOT / "tests" / "golden" / OUT, mean=b.mean(axis=0).astype(np.float32),
                    var_of_mean=(b.var(axis=0, ddof=1) / K).astype(np.float32), histories_per_run=NB * HPT, runs=K,
                    seeds=np.array([1000 + 7919 * k for k in range(K)]))
print("total detected energy/history per class:", runs.mean(axis=0).sum(axis=(1, 2)))
