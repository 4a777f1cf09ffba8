// decimal8.hpp -- the value a consumer of the reference's ASCII projection files ends up with, computed exactly.
//
// The reference prints every pixel as "%.8lf" (report_image, docker/mcgpu/MC-GPU_v1.3.cu:2886-2894); the Python side reads
// the text with np.loadtxt(float64) and converts to float32 (cbctmc/mc/projection.py:42-43).  decimal8_to_float(v) returns
// exactly that float32 without the text detour: v's exact binary value is rounded to 8 decimals (round-half-even, what
// printf does), the decimal is rounded to the nearest double (what strtod does: K / 1e8 with K < 2^53 is a correctly
// rounded division) and that double to the nearest float.  Integer arithmetic only for the decimal step, so host and
// device agree bit for bit.  Valid for 0 <= v < 4e7 (pixel values are eV/cm^2 per history: < 1e6).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define MCGPU_HD __host__ __device__
#else
#define MCGPU_HD
#endif

namespace mcgpu {

// 64 x 64 -> 128-bit product
MCGPU_HD inline void mul_64x64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
  const uint64_t a0 = a & 0xffffffffull, a1 = a >> 32, b0 = b & 0xffffffffull, b1 = b >> 32;
  const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const uint64_t mid = (p00 >> 32) + (p01 & 0xffffffffull) + (p10 & 0xffffffffull);
  lo = (p00 & 0xffffffffull) | (mid << 32);
  hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

// round(v * 1e8) with ties to even, on the exact value of v
MCGPU_HD inline uint64_t decimal8_digits(double v) {
  uint64_t bits;
  memcpy(&bits, &v, 8);
  const int be = (int)((bits >> 52) & 0x7ff);
  uint64_t m = bits & 0x000fffffffffffffull;
  int e;
  if (be == 0) e = -1074;  // subnormal (or zero)
  else { m |= 0x0010000000000000ull; e = be - 1075; }
  if (m == 0) return 0;
  uint64_t hi, lo;
  mul_64x64(m, 100000000ull, hi, lo);  // P = m * 1e8 < 2^80;  v * 1e8 = P * 2^e
  if (e >= 0) return lo << e;          // not reached for v < 4e7
  const int s = -e;                    // >= 27 for v < 4e7
  if (s > 81) return 0;                // P * 2^-s < 2^-1
  // integer part and remainder of P / 2^s
  uint64_t ip, rem_hi, rem_lo, half_hi, half_lo;
  if (s >= 64) {
    const int t = s - 64;  // 0..17
    ip = t == 0 ? hi : (hi >> t);
    rem_hi = t == 0 ? 0 : (hi & ((1ull << t) - 1));
    rem_lo = lo;
    half_hi = t == 0 ? 0 : (1ull << (t - 1));
    half_lo = t == 0 ? (1ull << 63) : 0;
  } else {
    ip = (hi << (64 - s)) | (lo >> s);  // s in 27..63; hi < 2^16 so nothing is lost
    rem_hi = 0;
    rem_lo = lo & ((1ull << s) - 1);
    half_hi = 0;
    half_lo = 1ull << (s - 1);
  }
  const bool above = (rem_hi > half_hi) || (rem_hi == half_hi && rem_lo > half_lo);
  const bool tie = (rem_hi == half_hi) && (rem_lo == half_lo);
  if (above || (tie && (ip & 1ull))) ++ip;
  return ip;
}

MCGPU_HD inline float decimal8_to_float(double v) { return (float)((double)decimal8_digits(v) / 100000000.0); }

// One pixel of the post-processed projection (projection.py:42-51, :118-127): the four float32 class values, their sums in
// numpy's order ((a0 + a1) + a2) + a3 and (a1 + a2) + a3.  norm = (1/100) * inv_px_X * inv_px_Z / N (MC-GPU_v1.3.cu:2860-2861).
MCGPU_HD inline void finalize_pixel(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3, double norm, float& total, float& unscattered,
                                    float& scattered) {
  const float f0 = decimal8_to_float(norm * (double)t0), f1 = decimal8_to_float(norm * (double)t1),
              f2 = decimal8_to_float(norm * (double)t2), f3 = decimal8_to_float(norm * (double)t3);
  total = ((f0 + f1) + f2) + f3;
  unscattered = f0;
  scattered = (f1 + f2) + f3;
}

}  // namespace mcgpu
