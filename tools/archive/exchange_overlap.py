"""Does the tally exchange disturb tracking?  (GPU, one process, one device)
A real rank (tracks 1e8 histories per step, never owns a step: policy rank0, it is rank 1) next to a stand-in owner (rank 0:
zero-history launches, collects).  Compares the tracking kernel's time with the exchange (45 MB copy-engine push per step beside
the next kernel + the owner's fused add) against plain launches into one buffer.
usage: exchange_overlap.py [workload dir] [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch, cases
eng = cases.pkg.engine
inp = (sys.argv[1] if len(sys.argv) > 1 else "/tmp/mcgpu_bench_catphan_512_894") + "/input.in"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
H = int(float(os.environ.get("H", "1e8")))
ctx = eng.create(inp, device=0)
owner_ctx = ctx.clone(0)
nz, nx = ctx.detector_shape
stream = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
seed, nproj = ctx.geti("seed"), ctx.num_projections
def plain():
    ms = []
    for i in range(steps):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch((i * 149) % nproj, image.data_ptr(), H, mode="fast", seed=seed, stream=stream)
        ms.append(ctx.last_kernel_ms())
    return ms[4:]
shared = bytearray(eng.Exchange.shared_bytes(2))
xs = [eng.Exchange(0, r, 2, ctx.image_words, shared, eng.EXCHANGE_ROOT0 | eng.EXCHANGE_LOCAL) for r in range(2)]
xs[0].connect_local(xs[1]); xs[1].connect_local(xs[0])
def exchanged():
    ms, push, add = [], [], []
    base = exchanged.k
    for i in range(steps):
        k = base + i
        t0 = xs[0].begin(k, side.cuda_stream)
        owner_ctx.launch(0, t0, 0, mode="fast", seed=seed, stream=side.cuda_stream)  # the stand-in owner tracks nothing
        xs[0].submit(k, side.cuda_stream)
        t1 = xs[1].begin(k, stream)
        ctx.launch((i * 149) % nproj, t1, H, mode="fast", seed=seed, stream=stream)
        xs[1].submit(k, stream)
        if k > 0:
            xs[0].collect(k - 1, side.cuda_stream)
            xs[1].collect(k - 1, stream)
        ms.append(ctx.last_kernel_ms())
        if i > 4:
            push.append(xs[1].stats()["last_push_ms"]); add.append(xs[0].stats()["last_add_ms"])
    exchanged.k = base + steps
    return ms[4:], push, add
exchanged.k = 0
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
src = torch.ones((4, nz, nx), dtype=torch.int64, device="cuda"); dst = torch.zeros_like(src)
def copy_beside(kind):
    """one 45 MB device-to-device copy of `kind` (3 = a kernel, 1024 = copy engine) issued 0.5 ms into each tracking launch"""
    ms, cp = [], []
    for i in range(steps):
        ctx.clear(image.data_ptr(), stream)
        ctx.launch((i * 149) % nproj, image.data_ptr(), H, mode="fast", seed=seed, stream=stream)
        time.sleep(0.0005)
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), src.numel() * 8, kind, side.cuda_stream)
        hip.hipStreamSynchronize(side.cuda_stream)
        cp.append((time.perf_counter() - t0) * 1e3)
        ms.append(ctx.last_kernel_ms())
    return float(np.mean(ms[4:])), float(np.mean(cp[4:]))
a = plain(); (b, push, add) = exchanged(); c = plain(); (d, push2, add2) = exchanged()
torch.cuda.synchronize()
k_sdma, c_sdma = copy_beside(1024); k_blit, c_blit = copy_beside(3); e = plain()
out = {"histories_per_step": H, "kernel_ms_plain": [float(np.mean(a)), float(np.mean(c))], "kernel_ms_with_exchange": [float(np.mean(b)), float(np.mean(d))],
       "push_ms_beside_the_next_kernel": float(np.mean(push + push2)), "push_GBps": ctx.image_words * 8 / (float(np.mean(push + push2)) * 1e-3) / 1e9,
       "fused_add_ms_one_peer_starved_beside_the_tracking_kernel_of_the_same_device": float(np.mean(add + add2)), "steps_per_series": steps - 4,
       "copy_engine_45MB_beside_tracking": {"kernel_ms": k_sdma, "copy_ms_host_timed": c_sdma, "GBps": src.numel() * 8 / (c_sdma * 1e-3) / 1e9},
       "blit_kernel_45MB_beside_tracking": {"kernel_ms": k_blit, "copy_ms_host_timed": c_blit}, "kernel_ms_plain_again": float(np.mean(e))}
print(json.dumps(out))
for x in xs: x.close()
owner_ctx.close(); ctx.close()
