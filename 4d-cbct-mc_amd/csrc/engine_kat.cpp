// engine_kat.cpp -- C ABI of the device-side known-answer hooks of the parity tests and of the micro-benchmarks behind the
// bench line's ceilings (include/mcgpu_amd.h: mcgpu_kat_*, mcgpu_microbench).
#include "engine_internal.hpp"

using namespace mcgpu;

extern "C" {

int mcgpu_microbench(mcgpu_ctx* ctx, int kind, double* out, int n_out) {
  ABI_BEGIN
  require(ctx && ctx->has_device && out && ((kind == MCGPU_MICROBENCH_VALU_ISSUE && n_out >= 3) || (kind == MCGPU_MICROBENCH_ATOMIC_RATE && n_out >= 1)), -1,
          "!!ERROR!! mcgpu_microbench: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  HIP_TRY(hipDeviceSynchronize());
  if (kind == MCGPU_MICROBENCH_VALU_ISSUE) HIP_TRY(microbench_valu_issue(ctx->dev.num_cus, out, nullptr));
  else HIP_TRY(microbench_atomic_rate(out, nullptr));
  return 0;
  ABI_END
}

int mcgpu_kat_rng(mcgpu_ctx* ctx, int mode, int seed, int batch, int hpt, int n, float* out_f32) {
  ABI_BEGIN
  require(ctx && ctx->has_device && out_f32 && n > 0, -1, "!!ERROR!! mcgpu_kat_rng: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  float* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, (size_t)n * 4));
  hipError_t e = launch_kat_rng(mode, seed, batch, hpt, n, d, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_f32, d, (size_t)n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_rng_streams(mcgpu_ctx* ctx, int generator, unsigned int seed, unsigned int projection, unsigned long long first_id,
                          const unsigned long long* ids, int n_ids, int n_draws, uint32_t* out_u32) {
  ABI_BEGIN
  require(ctx && ctx->has_device && out_u32 && n_ids > 0 && n_draws > 0 && (generator == 0 || generator == 1) &&
              (size_t)n_ids * (size_t)n_draws <= ((size_t)1 << 30),
          -1, "!!ERROR!! mcgpu_kat_rng_streams: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  unsigned int* d = nullptr;
  unsigned long long* d_ids = nullptr;
  const size_t nb = (size_t)n_ids * (size_t)n_draws * 4;
  HIP_TRY(hipMalloc((void**)&d, nb));
  hipError_t e = hipSuccess;
  if (ids) {
    e = hipMalloc((void**)&d_ids, (size_t)n_ids * 8);
    if (e == hipSuccess) e = hipMemcpy(d_ids, ids, (size_t)n_ids * 8, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = launch_kat_streams_fast(generator, seed, projection, first_id, d_ids, n_ids, n_draws, d, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_u32, d, nb, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (d_ids) (void)hipFree(d_ids);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_math(mcgpu_ctx* ctx, int n, const double* x, double* out_log, double* out_exp, double* out_sin, double* out_cos) {
  ABI_BEGIN
  require(ctx && ctx->has_device && x && out_log && out_exp && out_sin && out_cos && n > 0, -1, "!!ERROR!! mcgpu_kat_math: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  double* d = nullptr;
  const size_t nb = (size_t)n * 8;
  HIP_TRY(hipMalloc((void**)&d, 5 * nb));
  hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_kat_math(n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_log, d + n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_exp, d + 2 * n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_sin, d + 3 * n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_cos, d + 4 * n, nb, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_fast64(mcgpu_ctx* ctx, int n, const uint32_t* u, const double* a, const double* b, const double* c, const float* dir3, double* out8) {
  ABI_BEGIN
  require(ctx && ctx->has_device && u && a && b && c && dir3 && out8 && n > 0, -1, "!!ERROR!! mcgpu_kat_fast64: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  unsigned char* d = nullptr;
  const size_t n8 = (size_t)n * 8, n4 = (size_t)n * 4;
  HIP_TRY(hipMalloc((void**)&d, 11 * n8 + n4 + 3 * n4));  // out (8 n doubles) | a | b | c | u | dir
  double* d_out = (double*)d;
  double *d_a = d_out + 8 * (size_t)n, *d_b = d_a + n, *d_c = d_b + n;
  unsigned int* d_u = (unsigned int*)(d_c + n);
  float* d_dir = (float*)(d_u + n);
  hipError_t e = hipMemcpy(d_a, a, n8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_b, b, n8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_c, c, n8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_u, u, n4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_dir, dir3, 3 * n4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_kat_fast64(n, d_u, d_a, d_b, d_c, d_dir, d_out, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out8, d_out, 8 * n8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_f32(mcgpu_ctx* ctx, int op, int n, const float* a, const float* b, float* inout) {
  ABI_BEGIN
  require(ctx && ctx->has_device && a && b && inout && n > 0 && op >= 0 && op <= 4, -1, "!!ERROR!! mcgpu_kat_f32: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  float* d = nullptr;
  const size_t nb = (size_t)n * 4;
  HIP_TRY(hipMalloc((void**)&d, 3 * nb));
  hipError_t e = hipMemcpy(d, a, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + n, b, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)n, inout, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_kat_f32(op, n, d, d + n, d + 2 * (size_t)n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(inout, d + 2 * (size_t)n, nb, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_expf(mcgpu_ctx* ctx, int n, const float* x, float* out_exp) {
  ABI_BEGIN
  require(ctx && ctx->has_device && x && out_exp && n > 0, -1, "!!ERROR!! mcgpu_kat_expf: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  float* d = nullptr;
  const size_t nb = (size_t)n * 4;
  HIP_TRY(hipMalloc((void**)&d, 2 * nb));
  hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_kat_expf(n, d, d + n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_exp, d + n, nb, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return 0;
  ABI_END
}

int mcgpu_kat_tile_records(int n_tiles, const short* indices, uint32_t* out_u32) {
  ABI_BEGIN
  require(n_tiles > 0 && indices && out_u32, -1, "!!ERROR!! mcgpu_kat_tile_records: bad argument");
  for (int t = 0; t < n_tiles; ++t) {
    const TileRecord r = encode_tile_record(indices + (size_t)t * 64);
    out_u32[4 * t + 0] = r.ab;
    out_u32[4 * t + 1] = r.code;
    out_u32[4 * t + 2] = (uint32_t)r.mask;
    out_u32[4 * t + 3] = (uint32_t)(r.mask >> 32);
  }
  return 0;
  ABI_END
}

}  // extern "C"
