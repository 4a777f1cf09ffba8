"""What does work on a second stream cost the persistent tracking kernel?  (scan driver / RCCL-overlap design input)
For each variant: 12 projections; a side operation is issued on another stream `delay_us` after the launch."""
import sys, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch, cases
eng = cases.pkg.engine
ctx = eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0)
nz, nx = ctx.detector_shape
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream().cuda_stream
dev_small = torch.zeros(3 * nz * nx, dtype=torch.float32, device="cuda")
host_small = torch.zeros(3 * nz * nx, dtype=torch.float32).pin_memory()
dev_big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")  # 256 MB
dev_big2 = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
H = int(1e8)

def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass

def variant(name, op, delay_us):
    ms = []
    for i in range(12):
        ctx.clear(image.data_ptr(), main)
        ctx.launch((i * 149) % ctx.num_projections, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=main)
        if op is not None:
            spin(delay_us)
            with torch.cuda.stream(side):
                op()
        ms.append(ctx.last_kernel_ms())
        torch.cuda.synchronize()
    print(f"{name:34s} delay {delay_us:4d} us: kernel ms mean {np.mean(ms[2:]):.3f} min {np.min(ms[2:]):.3f} max {np.max(ms[2:]):.3f}", flush=True)

def before(name, op, n_ops=1):
    """side op issued on the other stream immediately BEFORE the tracking launch (running while the grid is dispatched)"""
    ms = []
    for i in range(12):
        ctx.clear(image.data_ptr(), main)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(n_ops):
                op()
        ctx.launch((i * 149) % ctx.num_projections, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=main)
        ms.append(ctx.last_kernel_ms())
        torch.cuda.synchronize()
    print(f"BEFORE {name:27s} x{n_ops}: kernel ms mean {np.mean(ms[2:]):.3f} min {np.min(ms[2:]):.3f} max {np.max(ms[2:]):.3f}", flush=True)

import os
for spare in (0, 25, 100):
    os.environ["MCGPU_GRID_SPARE_PERCENT"] = str(spare)
    ctx.reload_env_knobs()
    print("spare percent", spare)
    variant("alone", None, 0)
    before("256 MB add_", lambda: dev_big.add_(1.0))
    before("256 MB add_", lambda: dev_big.add_(1.0), 4)
ctx.close()
