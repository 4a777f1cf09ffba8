#!/usr/bin/env python3
"""Dev measurement (GPU): does a copy-engine (SDMA) transfer disturb the persistent tracking kernel?  A 45 MB device-to-pinned-host
copy (hipMemcpyAsync on a second stream: no compute unit involved, the same engines that carry peer-to-peer pushes over xGMI)
is issued (a) just BEFORE the tracking launch, so that it is in flight while the grid is dispatched, (b) 300 us AFTER it.
Compare with tools/overlap_probe.py, where a small KERNEL resident at dispatch time costs the tracking kernel up to 30 %."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import bench, cases
eng = cases.pkg.engine
wd = Path("/tmp/mcgpu_wl_catphan"); wd.mkdir(exist_ok=True)
inp = wd / "input.in"
if not inp.exists():
    inp = bench.build_workload(wd, "catphan", int(1e8), 894, eng)
torch.cuda.set_device(0)
ctx = eng.create(inp, device=0)
nz, nx = ctx.detector_shape
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
other = torch.ones((4, nz, nx), dtype=torch.int64, device="cuda")
host = torch.zeros((4, nz, nx), dtype=torch.int64).pin_memory()
side = torch.cuda.Stream()
main = torch.cuda.current_stream().cuda_stream
H = int(1e8)


def run(name, when, n_copies=1):
    ms, copy_ms = [], []
    for i in range(10):
        ctx.clear(image.data_ptr(), main)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def copies():
            with torch.cuda.stream(side):
                e0.record(side)
                for _ in range(n_copies):
                    host.copy_(other, non_blocking=True)
                e1.record(side)
        if when == "before":
            copies()
        ctx.launch((i * 149) % ctx.num_projections, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=main)
        if when == "after":
            t = time.perf_counter()
            while (time.perf_counter() - t) < 300e-6:
                pass
            copies()
        ms.append(ctx.last_kernel_ms())
        torch.cuda.synchronize()
        if when != "none":
            copy_ms.append(e0.elapsed_time(e1))
    print(f"{name:44s} kernel ms mean {np.mean(ms[2:]):.3f} (min {np.min(ms[2:]):.3f} max {np.max(ms[2:]):.3f})" +
          (f"; {n_copies} x 45 MB copy {np.mean(copy_ms[2:]):.3f} ms = {n_copies * 45.4 / np.mean(copy_ms[2:]):.1f} GB/s" if copy_ms else ""), flush=True)


run("alone", "none")
run("D2H copy in flight at dispatch", "before")
run("4 D2H copies in flight at dispatch", "before", 4)
run("D2H copy issued 300 us after the launch", "after")
run("4 D2H copies issued 300 us after the launch", "after", 4)
run("alone again", "none")
ctx.close()
