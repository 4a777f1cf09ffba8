// track_fast.hip -- FAST (production) personality of the photon-history kernel.
#define MC_COMPAT 0
#include "track_pool.inc"

namespace mcgpu {
namespace {
__global__ void kat_fast_rng(int seed, int hist, int n, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Rng r;
  rng_init_history(r, (unsigned long long)hist, (unsigned int)seed, 0u);
  for (int i = 0; i < n; ++i) out[i] = rng_f(r);
}
}  // namespace
hipError_t launch_kat_rng_fast(int seed, int hist, int n, float* out_dev, hipStream_t stream) {
  hipLaunchKernelGGL(kat_fast_rng, dim3(1), dim3(64), 0, stream, seed, hist, n, out_dev);
  return hipGetLastError();
}
}  // namespace mcgpu
