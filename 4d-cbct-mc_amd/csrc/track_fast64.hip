// track_fast64.hip -- the FAST personality with the reference's double-precision sub-steps (MCGPU_MODE_FAST_F64):
// rotate_double (K.cu:1103-1148), GRAa (K.cu:1181-1246) and GCOa's cdt1 / costh chain (K.cu:1329-1331) in double, as the
// reference computes them; scheduling, random-number streams and everything the reference does in float: track_fast.hip's.
#define MC_COMPAT 0
#define MC_FAST_F64 1
#include "track_pool.inc"

// Known-answer hook of this personality's double-precision helpers (mcgpu_kat_fast64; tests/test_gpu_parity.py): the statistical
// tests cannot see an azimuth that is wrong by a quarter turn -- scattering is symmetric about the photon's direction -- so
// sincos_turn, rsqrt_d, sqrt_ratio_d, compton_cdt1 and the rotation itself are compared with double-precision numpy directly.
//   out[i] = {sin, cos of 2 pi (u[i] + 1/2) 2^-32,  1 / sqrt(a[i]),  sqrt(a[i] / b[i]),  cdt1(tau = (float)a[i], E = (float)b[i] 1e5),
//             direction (u, v, w) of dir[i] rotated by the polar cosine c[i] and the azimuth of u[i]}   (8 doubles per item)
namespace mcgpu {
namespace {
__global__ void kat_fast64_kernel(int n, const unsigned int* u, const double* a, const double* b, const double* c, const float* dir, double* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  double s, co;
  sincos_turn(u[i], s, co);
  Particle P;
  P.x = P.y = P.z = 0.f; P.E = 0.f;
  P.u = dir[3 * i]; P.v = dir[3 * i + 1]; P.w = dir[3 * i + 2];
  Rng r;
  // one multiply-with-carry step must produce u[i]: x' = lo(a x + c) with x = 0 gives x' = c
  r.x = 0u; r.c = u[i];
  rotate_dir(P, c[i], r);
  double* o = out + 8 * (size_t)i;
  o[0] = s; o[1] = co; o[2] = rsqrt_d(a[i]); o[3] = sqrt_ratio_d(a[i], b[i]); o[4] = compton_cdt1((float)a[i], (float)b[i] * 1.0e5f);
  o[5] = (double)P.u; o[6] = (double)P.v; o[7] = (double)P.w;
}
}  // namespace
hipError_t launch_kat_fast64(int n, const unsigned int* u, const double* a, const double* b, const double* c, const float* dir, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(kat_fast64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, u, a, b, c, dir, out);
  return hipGetLastError();
}
}  // namespace mcgpu
