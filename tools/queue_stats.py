"""Diagnostics of the event-queue kernel on the bench workload (GPU)."""
import os, sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, cases
eng = cases.pkg.engine
os.environ["MCGPU_FAST_KERNEL"] = "queue"; os.environ["MCGPU_QUEUE_STATS"] = "1"
ctx = eng.create(os.environ.get("TUNE_INPUT", "/tmp/mcgpu_bench_512_894/input.in"), device=0)
nz, nx = ctx.detector_shape
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
H = int(float(os.environ.get("TUNE_HIST", "1e8")))
out = (C.c_ulonglong * 28)()
for rep in range(2):
    ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, 28, 1)
    ctx.clear(image.data_ptr(), stream)
    ctx.launch(0, image.data_ptr(), H, mode="fast", seed=1, first=0, stream=stream)
    ms = ctx.last_kernel_ms()
ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, 28, 0)
q = [int(v) for v in out]
names = ["FLIGHT", "NEW", "CFEW", "CMANY", "RAYLEIGH"]
print(f"kernel {ms:.2f} ms")
for k, n in enumerate(names):
    print(f"{n:9s} batches/hist {q[2*k]/H:.5f}  lanes/batch {q[2*k+1]/max(q[2*k],1):.1f}")
print("idle polls/hist", q[10] / H, " partial waits/hist", q[11] / H)
print("flight steps (wave) per hist", q[12] / H, " settle rounds per hist", q[13] / H)
tot = max(q[15], 1)
print("cycles: total/hist", q[15] / H, " stage %.3f pop %.3f push %.3f" % (q[14] / tot, q[16] / tot, q[17] / tot))
print("stage cycles/hist: flight %.1f new %.1f compton %.1f rayleigh %.1f" % tuple(q[18 + i] / H for i in range(4)))
ctx.close()
