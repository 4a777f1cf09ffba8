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

// Known-answer / statistics hook of the per-history streams (mcgpu_kat_rng_streams): out[i * n_draws + k] = k-th raw 32-bit
// output of the stream of history ids[i] (or first_id + i) at projection `stream_key`.
//   generator 0: the production stream (rng_init_history + rng_u32: Philox4x32-7 seeding, multiply-with-carry steps)
//   generator 1: the yardstick of tests/test_fast_rng.py -- Philox4x32-10 evaluated PER DRAW (counter = {id, projection, k / 4},
//                word k % 4), a generator that passes BigCrush (Salmon et al., SC11) and has no state to correlate
__global__ void kat_fast_streams(int generator, unsigned int seed, unsigned int stream_key, unsigned long long first_id,
                                 const unsigned long long* ids, int n_ids, int n_draws, unsigned int* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n_ids) return;
  const unsigned long long id = ids ? ids[i] : first_id + (unsigned long long)i;
  unsigned int* o = out + (size_t)i * (size_t)n_draws;
  if (generator == 0) {
    Rng r;
    rng_init_history(r, id, seed, stream_key);
    for (int k = 0; k < n_draws; ++k) o[k] = rng_u32(r);
  } else {
    for (int k4 = 0; k4 < n_draws; k4 += 4) {
      unsigned int c0 = (unsigned int)id, c1 = (unsigned int)(id >> 32), c2 = stream_key, c3 = (unsigned int)(k4 >> 2);
      unsigned int k0 = seed, k1 = 0xCB435443u;
      for (int round = 0; round < 10; ++round) {
        const unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
      }
      const unsigned int w[4] = {c0, c1, c2, c3};
      for (int j = 0; j < 4 && k4 + j < n_draws; ++j) o[k4 + j] = w[j];
    }
  }
}
}  // namespace
hipError_t launch_kat_rng_fast(int seed, int hist, int n, float* out_dev, hipStream_t stream) {
  hipLaunchKernelGGL(kat_fast_rng, dim3(1), dim3(64), 0, stream, seed, hist, n, out_dev);
  return hipGetLastError();
}
hipError_t launch_kat_streams_fast(int generator, unsigned int seed, unsigned int stream_key, unsigned long long first_id,
                                   const unsigned long long* ids_dev, int n_ids, int n_draws, unsigned int* out_dev, hipStream_t stream) {
  hipLaunchKernelGGL(kat_fast_streams, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, stream, generator, seed, stream_key, first_id,
                     ids_dev, n_ids, n_draws, out_dev);
  return hipGetLastError();
}
}  // namespace mcgpu
