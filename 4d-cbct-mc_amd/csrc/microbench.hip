// microbench.hip -- the two hardware ceilings bench.py prices the FAST kernel against, measured in the run that reports them
// (mcgpu_microbench; about 20 ms each).  No reference counterpart: measurement support of SURVEY.md 8d.
//
//   kind 0  vector-instruction issue: a dense chain of dependent FMAs, 8 waves per SIMD (two 1024-thread workgroups per CU like
//           the tracking kernel), under three EXEC masks -- 64 active lanes, lanes 0-31 (two of the four 16-lane groups idle),
//           32 lanes spread over all groups.  out = wave-instructions per ns and SIMD for each.  The tracking kernel issues at
//           55-60 % lane utilisation, i.e. between the first two.
//   kind 1  scattered 64-bit atomic adds without return into a detector-sized tally (4 x 1848 x 768 words = 45 MB), the access
//           pattern of tally_image (MC-GPU_kernel_v1.3.cu:482-604).  out[0] = adds per second, chip-wide.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mcgpu {
namespace {

__global__ __launch_bounds__(1024) void fma_chain_kernel(float* out, int mode, int iters) {
  const int lane = threadIdx.x & 63;
  const bool active = mode == 0 ? true : (mode == 1 ? lane < 32 : (lane & 1) == 0);
  float a = (float)threadIdx.x * 1e-3f, b = 1.0000001f, c = 1e-7f;
  if (active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) a = fmaf(a, b, c);
    }
  }
  if (a == 123.456f) out[0] = a;  // never true: keeps the chain alive
}

__global__ void scatter_add_kernel(unsigned long long* img, unsigned int words, int per_thread, unsigned int seed) {
  unsigned int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    atomicAdd(img + (x % words), (unsigned long long)(x & 0xffffu));
  }
}

}  // namespace

// out3 = wave-instructions per ns and SIMD {64 lanes, lanes 0-31, 32 lanes spread}
hipError_t microbench_valu_issue(int num_cus, double out3[3], hipStream_t stream) {
  float* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, 4);
  if (e != hipSuccess) return e;
  hipEvent_t a = nullptr, b = nullptr;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int blocks = 2 * num_cus, iters = 12000;
  for (int mode = 0; mode < 3 && e == hipSuccess; ++mode) {
    hipLaunchKernelGGL(fma_chain_kernel, dim3(blocks), dim3(1024), 0, stream, d, mode, 500);  // warm: clocks, code
    (void)hipEventRecord(a, stream);
    hipLaunchKernelGGL(fma_chain_kernel, dim3(blocks), dim3(1024), 0, stream, d, mode, iters);
    (void)hipEventRecord(b, stream);
    e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 16.0 * (double)iters * 32.0;  // FMA wave-instructions (loop overhead: 2 scalar per 32)
    out3[mode] = ms > 0.f ? insts / (4.0 * num_cus) / ((double)ms * 1e6) : 0.0;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(d);
  return e;
}

// out[0] = scattered 64-bit atomic adds per second
hipError_t microbench_atomic_rate(double* out, hipStream_t stream) {
  const unsigned int words = 4u * 1848u * 768u;
  unsigned long long* img = nullptr;
  hipError_t e = hipMalloc((void**)&img, (size_t)words * 8);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(img, 0, (size_t)words * 8, stream);
  hipEvent_t a = nullptr, b = nullptr;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int blocks = 256 * 8, threads = 256, per = 512;
  float best = 0.f;
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {  // the first pass warms the caches and the clocks
    (void)hipEventRecord(a, stream);
    hipLaunchKernelGGL(scatter_add_kernel, dim3(blocks), dim3(threads), 0, stream, img, words, per, 17u + rep);
    (void)hipEventRecord(b, stream);
    e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && (best == 0.f || ms < best)) best = ms;
  }
  *out = best > 0.f ? (double)blocks * threads * per / ((double)best * 1e-3) : 0.0;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(img);
  return e;
}

}  // namespace mcgpu
