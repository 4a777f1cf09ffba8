"""Dev check (GPU): one launch per seed of the engine library in MCGPU_AMD_LIB (A/B builds), timed."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, cases
eng = cases.pkg.engine
wd = Path('/tmp/mcgpu_bench_thorax_512_894')
if not (wd / 'input.in').exists():
    wd.mkdir(parents=True, exist_ok=True); bench.build_workload(wd, 'thorax', int(1e8), 894, eng)
n = int(float(sys.argv[1])); seeds = [int(v) for v in sys.argv[2:]]
with eng.create(wd / 'input.in', device=0) as ctx:
    for seed in seeds:
        t0 = time.time(); img, s, d = ctx.run_projection(600, n, mode='fast', seed=seed)
        print(os.environ.get('MCGPU_AMD_LIB', 'default').rsplit('/', 1)[-1], 'seed', seed, n, 'kernel s', round(s, 4), 'rate', f'{d / s:.3e}', 'sum', int(img.sum()), flush=True)
