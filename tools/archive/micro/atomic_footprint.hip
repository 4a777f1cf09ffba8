// Micro-benchmark: rate of scattered integer atomic adds (no return) against the footprint of the tally and the word width.
// Question: is the tally's cost the atomic unit or the line traffic of a 45 MB image that no L2 (4 MB per XCD) can hold?
// Build: hipcc --offload-arch=gfx950 -O3 atomic_footprint.hip -o atomic_footprint
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void scatter_add(T* img, size_t words, int per_thread, unsigned seed) {
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const size_t idx = (size_t)(((unsigned long long)x * words) >> 32);
    atomicAdd(img + idx, (T)(x & 0xffff));
  }
}
template <typename T>
void run(const char* name, T* img, size_t words) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    const int blocks = 256 * 8, threads = 256, per = 256;
    hipEventRecord(a);
    hipLaunchKernelGGL(scatter_add<T>, dim3(blocks), dim3(threads), 0, 0, img, words, per, 17u + rep);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("%s footprint %8.2f MB: %.3e adds/s\n", name, words * sizeof(T) / 1048576.0, 256.0 * 8 * 256 * 256 / (best * 1e-3));
}
int main() {
  void* buf; hipMalloc(&buf, 256u << 20); hipMemset(buf, 0, 256u << 20);
  const double mb[] = {0.25, 1, 2, 4, 5.7, 8, 11.4, 16, 22.7, 32, 45.4, 64, 128};
  for (double m : mb) {
    run<unsigned long long>("u64", (unsigned long long*)buf, (size_t)(m * 1048576 / 8));
    run<unsigned int>("u32", (unsigned int*)buf, (size_t)(m * 1048576 / 4));
  }
  return 0;
}
