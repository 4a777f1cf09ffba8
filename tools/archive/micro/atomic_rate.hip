// Micro-benchmark: rate of scattered 64-bit integer atomic adds (no return) into a detector-sized tally (45 MB), the
// access pattern of tally_image.  Build: hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void scatter_add(unsigned long long* img, size_t words, int per_thread, unsigned seed, int mode) {
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    size_t idx = (size_t)x % words;
    if (mode == 1) idx = (idx % (words / 4));         // one class only (primaries)
    if (mode == 2) img[idx] += 1;                      // plain RMW (wrong, for comparison)
    else if (mode == 3) {                              // one private tally per XCD, atomics of workgroup scope (performed in that XCD's L2)
      const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11)) & 7u;
      __hip_atomic_fetch_add(img + (size_t)xcc * words + idx, (unsigned long long)(x & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else atomicAdd(img + idx, (unsigned long long)(x & 0xffff));
  }
}
int main() {
  const size_t words = 4ull * 1848 * 768;
  unsigned long long* img;
  hipMalloc(&img, 8 * words * 8);  // 8 copies for mode 3
  hipMemset(img, 0, 8 * words * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      const int blocks = 256 * 8, threads = 256, per = 512;
      hipEventRecord(a);
      hipLaunchKernelGGL(scatter_add, dim3(blocks), dim3(threads), 0, 0, img, words, per, 17u + rep, mode);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d: %.3e scattered 64-bit adds/s (%.2f ms)\n", mode, (double)blocks * threads * per / (ms * 1e-3), ms);
    }
  // mode 3 check: the 8 private copies must add up to what the launches put in (no lost updates)
  {
    unsigned long long* h = (unsigned long long*)malloc(8 * words * 8);
    hipMemcpy(h, img, 8 * words * 8, hipMemcpyDeviceToHost);
    unsigned long long sum_priv = 0, sum_shared_part = 0;
    for (size_t i = 0; i < 8 * words; ++i) sum_priv += h[i];
    printf("sum over all copies %llu\n", sum_priv);
    free(h);
  }
  return 0;
}
