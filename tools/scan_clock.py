#!/usr/bin/env python3
"""Dev measurement (GPU): why a kernel of the 894-projection scan takes longer than a kernel of the headline's six launches.
Back-to-back FAST launches of 1e8 histories with nothing beside them (no finalize, no copies, no writer): kernel time by HIP events
for launches 1-6 and for launches 895-900, and the shader clock the waves actually ran at (diagnostic build: per-wave s_memtime
cycles over s_memrealtime ticks of 100 MHz) after 6 and after 900 launches.  Usage: python tools/scan_clock.py [workload]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("MCGPU_AMD_LIB", str(ROOT / "4d-cbct-mc_amd" / "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import torch
import bench, cases
eng = cases.pkg.engine
NS, NT = 32, 32 + 3 * 16384
wl = sys.argv[1] if len(sys.argv) > 1 else "catphan"
wd = Path(f"/tmp/mcgpu_bench_{wl}_512_894")
if not (wd / "input.in").exists():
    wd.mkdir(parents=True, exist_ok=True)
    bench.build_workload(wd, wl, int(1e8), 894, eng)


def clock(ctx):
    out = (C.c_ulonglong * NT)()
    ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 1)
    ctx.run_projection(7, int(1e8), mode="stats", seed=1)
    ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 0)
    a = np.frombuffer(out, dtype=np.uint64)[NS:].reshape(-1, 3)
    a = a[a[:, 1] != 0]
    cyc = ((a[:, 0] >> np.uint64(36)).astype(np.float64)) * 256.0
    ticks = (a[:, 2].astype(np.int64) - a[:, 1].astype(np.int64)).astype(np.float64)
    ok = ticks > 1000
    return float(np.median(cyc[ok] / ticks[ok] * 100.0))


with eng.create(wd / "input.in", device=0) as ctx:
    nz, nx = ctx.detector_shape
    image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed = ctx.geti("seed")

    def launches(n, first):
        ev = []
        for i in range(n):
            ctx.clear(image.data_ptr(), stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ctx.launch(((first + i) * 149) % ctx.num_projections, image.data_ptr(), int(1e8), mode="fast", seed=seed, stream=stream)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    launches(2, 0)  # one-off costs
    torch.cuda.synchronize()
    import time
    time.sleep(2.0)  # an idle GPU, like the one the headline region starts on
    first6 = launches(6, 2)
    mhz6 = clock(ctx)
    time.sleep(2.0)
    all900 = launches(900, 8)
    mhz900 = clock(ctx)
    print(f"{wl}: kernel ms, launches 1-6 after idling: mean {np.mean(first6):.3f} (each {np.round(first6, 3).tolist()}); shader clock right after: {mhz6:.0f} MHz")
    print(f"{wl}: kernel ms of 900 back-to-back launches: 1-6 {np.mean(all900[:6]):.3f}, 7-100 {np.mean(all900[6:100]):.3f}, 101-500 {np.mean(all900[100:500]):.3f}, "
          f"501-894 {np.mean(all900[500:894]):.3f}, 895-900 {np.mean(all900[894:]):.3f}; shader clock right after: {mhz900:.0f} MHz")
