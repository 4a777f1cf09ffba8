#!/usr/bin/env python3
"""Dev check (GPU): FAST (MODE=fast64: the variant with the reference's double-precision sub-steps) against the bit-exact COMPAT personality on a bench workload with many histories on both sides.
K independent runs per mode (different seeds); the run-to-run scatter gives the variances.  Per scatter class: detected energy
per history, ratio FAST / COMPAT and its z; per 32x32-pixel block: z from the same run-to-run variances.
Usage: python tools/fast_vs_compat.py <workload> <projection> [runs] [histories per run]"""
import sys, os, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
wl, p = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n = int(float(sys.argv[4])) if len(sys.argv) > 4 else 500_000_000
eng = cases.pkg.engine
MODE = os.environ.get("MODE", "fast")
inp = cases.pkg.workloads.workload_dir(wl) / "input.in"
if not inp.exists():  # a fresh box: the workload's inputs in the reference's wire formats (4d-cbct-mc_amd/workloads.py)
    cases.pkg.workloads.build_workload(inp.parent, wl, 100_000_000, 894, engine=eng)
B = 32
def blocks(img):
    c, nz, nx = img.shape
    return img[:, :nz // B * B, :nx // B * B].reshape(c, nz // B, B, nx // B, B).sum(axis=(2, 4)).astype(np.float64)
with eng.create(inp, device=0) as ctx:
    batches, hpt, _ = ctx.reference_shape(n)
    per = {"fast": [], "compat": []}
    t0 = time.time()
    for k in range(K):
        img, _, d = ctx.run_projection(p, n, mode=MODE, seed=int(os.environ.get("SEED0", "1000")) + k)
        per["fast"].append(blocks(img) / d)
        img, _, d = ctx.run_projection(p, batches, mode="compat", seed=int(os.environ.get("SEED0", "1000")) + 1000 + 7 * k, hpt=hpt)
        per["compat"].append(blocks(img) / d)
    print(f"{wl} projection {p}, mode {MODE}: {K} runs of {n:.2e} histories per mode in {time.time() - t0:.1f} s")
    F, Cc = np.array(per["fast"]), np.array(per["compat"])          # [K, 4, bz, bx]
    ef, ec = F.sum(axis=(2, 3)), Cc.sum(axis=(2, 3))                 # [K, 4] energy per history per class
    for c, name in enumerate(("primary", "compton", "rayleigh", "multiple")):
        mf, mc = ef[:, c].mean(), ec[:, c].mean()
        se = np.sqrt(ef[:, c].var(ddof=1) / K + ec[:, c].var(ddof=1) / K)
        print(f"  {name:9s} FAST/COMPAT {mf / mc:.6f}  z {(mf - mc) / se:+.2f}  (relative sigma {se / mc:.1e})")
    mfb, mcb = F.mean(axis=0), Cc.mean(axis=0)
    seb = np.sqrt(F.var(axis=0, ddof=1) / K + Cc.var(axis=0, ddof=1) / K)
    for c, name in enumerate(("primary", "compton", "rayleigh", "multiple")):
        m = (mcb[c] > 0) & (seb[c] > 0)
        z = (mfb[c][m] - mcb[c][m]) / seb[c][m]
        print(f"  {name:9s} {z.size} blocks of {B}x{B}: mean z {z.mean():+.3f}, std {z.std():.3f} (Student t, {2 * K - 2} dof: {np.sqrt((2 * K - 2) / (2 * K - 4)):.3f} expected), beyond 4: {int((np.abs(z) > 4).sum())}")
        if c == 0:
            # the primary beam ends at detector column 1024 (the half-fan crop of the reference, proj.py:42-51): the handful of
            # photons within 0.01 pixel of that edge fall on either side of it depending on the last bits of the direction
            edge = 1024 // B
            keep = m.copy(); keep[:, edge] = False
            ze = (mfb[c][keep] - mcb[c][keep]) / seb[c][keep]
            frac = mcb[c][:, edge].sum() / mcb[c].sum()
            print(f"            without the block column of the beam edge (x = {edge * B}..{edge * B + B - 1}, {frac:.1e} of the primary energy): "
                  f"mean z {ze.mean():+.3f}, std {ze.std():.3f}, beyond 4: {int((np.abs(ze) > 4).sum())}")
    if os.environ.get("ZMAP"):
        m = (mcb[0] > 0) & (seb[0] > 0)
        z = np.where(m, (mfb[0] - mcb[0]) / np.where(seb[0] > 0, seb[0], 1), 0)
        print("primary z map (rows = detector z blocks, columns = x blocks), rounded:")
        for row in z:
            print(" ".join(f"{int(round(v)):+d}" if abs(v) >= 2.5 else " ." for v in row))
        print("relative difference of the blocks beyond 3.5:", [(int(i), int(j), round(float((mfb[0][i, j] - mcb[0][i, j]) / mcb[0][i, j]), 6)) for i, j in zip(*np.where(np.abs(z) > 3.5))][:40])
        cols = (mfb[0].sum(axis=0) - mcb[0].sum(axis=0)) / np.sqrt((seb[0] ** 2).sum(axis=0))
        rows = (mfb[0].sum(axis=1) - mcb[0].sum(axis=1)) / np.sqrt((seb[0] ** 2).sum(axis=1))
        print("column z:", np.round(cols, 1).tolist())
        print("row z:", np.round(rows, 1).tolist())
