"""Where do the persistent tracking kernel's waves land, alone vs. launched while another kernel is running?
Diagnostic (stats) build: per wave {HW_ID, XCC_ID, first/last clock (100 MHz)}; see track_pool.inc."""
import sys, ctypes as C, collections
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
__import__("os").environ.setdefault("MCGPU_AMD_LIB", __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "4d-cbct-mc_amd", "libmcgpu_amd_stats.so"))  # the diagnostic build (stats mode)
import numpy as np, torch, cases
eng = cases.pkg.engine
ctx = eng.create("/tmp/mcgpu_bench_catphan_512_894/input.in", device=0)
nz, nx = ctx.detector_shape
image = torch.zeros((4, nz, nx), dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream().cuda_stream
dev_big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
H = int(1e8)
NT = 28 + 3 * 16384

def run(name, interfere):
    for rep in range(3):
        out = (C.c_ulonglong * NT)()
        ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 1)
        ctx.clear(image.data_ptr(), main)
        torch.cuda.synchronize()
        if interfere:
            with torch.cuda.stream(side):
                dev_big.add_(1.0)
        ctx.launch(7, image.data_ptr(), H, mode="stats", seed=1, first=0, stream=main)
        ms = ctx.last_kernel_ms()
        torch.cuda.synchronize()
        ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, NT, 0)
    a = np.frombuffer(out, dtype=np.uint64)[28:].reshape(-1, 3)
    a = a[a[:, 1] != 0]
    hw = (a[:, 0] & 0xffffffff).astype(np.int64); xcc = (a[:, 0] >> 32).astype(np.int64) & 0xf
    wave, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    t0 = a[:, 1].astype(np.int64); t1 = a[:, 2].astype(np.int64)
    base = t0.min()
    print(f"== {name}: kernel {ms:.2f} ms, waves traced {len(a)}")
    per_cu = collections.Counter(zip(xcc, se, sh, cu))
    per_simd = collections.Counter(zip(xcc, se, sh, cu, simd))
    print("   distinct CUs", len(per_cu), "waves/CU histogram", sorted(collections.Counter(per_cu.values()).items()))
    print("   waves/SIMD histogram", sorted(collections.Counter(per_simd.values()).items()))
    print("   waves/XCC", sorted(collections.Counter(xcc).items()))
    st = (t0 - base) / 100.0  # us
    en = (t1 - base) / 100.0
    print(f"   wave start us: p50 {np.percentile(st,50):.1f} p90 {np.percentile(st,90):.1f} p99 {np.percentile(st,99):.1f} max {st.max():.1f}; "
          f"end us: min {en[en>0].min():.0f} p50 {np.percentile(en[en>0],50):.0f} max {en.max():.0f}")

run("alone", False)
run("launched under a running 256 MB add_", True)
run("alone again", False)
ctx.close()
