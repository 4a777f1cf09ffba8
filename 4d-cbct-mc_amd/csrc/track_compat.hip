// track_compat.hip -- COMPAT personality of the photon-history kernel (bit-exact against the CPU oracle).
// Build with -ffp-contract=off: no fused multiply-add may be formed, every float/double op rounds once.
#define MC_COMPAT 1
#include "track_kernel.inc"

namespace mcgpu {
namespace {
// Known-answer kernels (mcgpu_kat_rng / mcgpu_kat_math in include/mcgpu_amd.h)
__global__ void kat_ranecu(int seed, int batch, int hpt, int n, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Rng r;
  rng_init_batch(r, (unsigned long long)batch, hpt, seed);
  for (int i = 0; i < n; ++i) out[i] = rng_f(r);
}
__global__ void kat_math(int n, const double* x, double* l, double* e, double* s, double* c) {
  stage_exp2_table();
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  l[i] = pm_log(x[i]);
  e[i] = pm_exp(x[i]);
  double sn, cs;
  pm_sincos(x[i], sn, cs);
  s[i] = sn;
  c[i] = cs;
}
__global__ void kat_expf(int n, const float* x, float* e) {
  stage_exp2_table();
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) e[i] = (x[i] <= 0.5f) ? mc_expf(x[i]) : gl_expf<false>(x[i]);  // mc_expf: the kernel's call (arguments <= 0.5 only)
}
// op 0: cm_sqrtf(a), 1: sqrtf(a) (the compiler's), 2: cm_divf(a, b), 3: a / b (the compiler's), 4: shell_pz(a, b, c) -- c in out on entry
__global__ void kat_f32(int op, int n, const float* a, const float* b, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  if (op == 0) r = cm_sqrtf(a[i]);
  else if (op == 1) r = sqrtf(a[i]);
  else if (op == 2) r = cm_divf(a[i], b[i]);
  else if (op == 3) r = a[i] / b[i];
  else r = shell_pz(a[i], b[i], out[i]);
  out[i] = r;
}
}  // namespace

hipError_t launch_kat_rng_fast(int seed, int hist, int n, float* out_dev, hipStream_t stream);

hipError_t launch_kat_rng(int mode, int seed, int batch, int hpt, int n, float* out_dev, hipStream_t stream) {
  if (mode == 1) {
    hipLaunchKernelGGL(kat_ranecu, dim3(1), dim3(64), 0, stream, seed, batch, hpt, n, out_dev);
    return hipGetLastError();
  }
  return launch_kat_rng_fast(seed, batch, n, out_dev, stream);
}
hipError_t launch_kat_math(int n, const double* x, double* l, double* e, double* s, double* c, hipStream_t stream) {
  hipLaunchKernelGGL(kat_math, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, l, e, s, c);
  return hipGetLastError();
}
hipError_t launch_kat_expf(int n, const float* x, float* e, hipStream_t stream) {
  hipLaunchKernelGGL(kat_expf, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, e);
  return hipGetLastError();
}
hipError_t launch_kat_f32(int op, int n, const float* a, const float* b, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(kat_f32, dim3((n + 255) / 256), dim3(256), 0, stream, op, n, a, b, out);
  return hipGetLastError();
}
}  // namespace mcgpu
