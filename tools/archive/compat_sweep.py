"""COMPAT personality: histories/s on a bench workload for a set of batching thresholds (GPU).
usage: compat_sweep.py <workload dir> "tC,tR,tN[,tTake]" ...   (MCGPU_COMPAT_THRESH_COMPTON / _RAYLEIGH / _NEW; tallies do not depend on them)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
KEYS = ("MCGPU_COMPAT_THRESH_COMPTON", "MCGPU_COMPAT_THRESH_RAYLEIGH", "MCGPU_COMPAT_THRESH_NEW", "MCGPU_COMPAT_THRESH_TAKE")
with eng.create(sys.argv[1] + "/input.in", device=0) as ctx:
    batches, hpt, total = ctx.reference_shape(int(float(os.environ.get("H", "2e7"))))
    ref = None
    for cfg in sys.argv[2:]:
        for k, v in zip(KEYS, cfg.split(",")):
            os.environ[k] = v
        ctx.reload_env_knobs()
        ctx.run_projection(300, batches // 8, mode="compat", seed=42, hpt=hpt)
        img, secs, done = ctx.run_projection(300, batches, mode="compat", seed=42, hpt=hpt)
        same = "" if ref is None else ("  tallies identical" if (img == ref).all() else "  TALLIES DIFFER")
        ref = img if ref is None else ref
        print(f"{cfg:12s} {done / secs / 1e9:.4f} e9 histories/s ({secs * 1e3:.1f} ms){same}", flush=True)
