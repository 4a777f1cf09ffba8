"""Kernel-time sweep of the FAST scheduler knobs on the bench workload (GPU).
usage: tune.py "tC,tR,tN,flyable_low,swap_batch" ...   (5 launches of 1e8 histories each, mean kernel ms)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, cases
eng = cases.pkg.engine
KEYS = ("MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW", "MCGPU_FLYABLE_LOW", "MCGPU_SWAP_BATCH")
ctx = eng.create(os.environ.get("TUNE_INPUT", "/tmp/mcgpu_bench_catphan_512_894/input.in"), device=0)
nz, nx = ctx.detector_shape
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
H = int(float(os.environ.get("TUNE_HIST", "1e8")))
res = []
for i in range(8):  # untimed: the first launches of a process run 4 % slower (clocks, caches) and would bias the first entry
    ctx.clear(image.data_ptr(), stream)
    ctx.launch((i * 149) % ctx.num_projections, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=stream)
    ctx.last_kernel_ms()
for cfg in sys.argv[1:]:
    for k in KEYS:
        os.environ.pop(k, None)
    for k, v in zip(KEYS, [x for x in cfg.split(",") if x]):
        os.environ[k] = v
    ctx.reload_env_knobs()
    ms = []
    for i in range(7):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch((i * 149) % ctx.num_projections, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=stream)
        ms.append(ctx.last_kernel_ms())
    res.append((float(np.mean(ms[2:])), cfg))
    print(f"{cfg:24s} {np.mean(ms[2:]):.3f} ms  (min {np.min(ms[2:]):.3f})", flush=True)
print("best:", sorted(res)[:5])
ctx.close()
