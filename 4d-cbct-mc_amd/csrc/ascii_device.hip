// ascii_device.hip -- the data lines of the reference's projection file, formatted on the device (row a16).
//
// report_image (docker/mcgpu/MC-GPU_v1.3.cu:2783-2953) prints four numbers per pixel with "%.8lf" -- 1.42 M lines, 63 MB of
// text per 1848 x 768 projection: ~0.9 s of single-threaded fprintf in the reference, 11-14 ms with 16 host threads here
// (report.cpp), against 4.5 ms for the tracking kernel.  Byte work with no reuse: 45 MB of tallies in, 63 MB of text out.
// Here the GPU writes the text itself; the host only moves bytes (copy-engine download, pwrite).
//
//   "%.8lf" of v = NORM * (double)e is the decimal expansion of the binary value rounded half-to-even at the 8th digit
//   (glibc is exact).  v = m * 2^x (m < 2^53), so v * 10^8 = (m * 10^8) * 2^x with the product < 2^80 held in 128 bits: the
//   shift, the remainder test against one half and the tie-to-even are exact integer operations -- no fallback path.
//
// Two passes over the tallies, one workgroup per detector row (the lines of a row are contiguous in the file), one pixel
// per thread and tile of 256 pixels (coalesced tally reads):
//   ascii_measure_kernel  line lengths -> bytes per row; row sums and the row's largest pixel for the footer
//   ascii_write_kernel    offset of the row (sum of the rows before it), then per tile: format, exclusive scan of the
//                         line lengths, characters into an LDS image of the tile's text, and the image copied out in
//                         aligned 16-byte stores (the image starts at the file offset's phase within 16 bytes)
// Round-2 history: the first version gave every thread a run of 7 pixels and let it store its characters one by one to
// global memory (182 us), found the row offsets with a single-thread scan kernel (109 us) and measured in 65 us.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ascii_device.hpp"

namespace mcgpu {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kIntegerDigits = 11;                          // values below 1e11 (engine.cpp: mcgpu_format_projection sizes the buffers for that)
constexpr unsigned long long kIntegerLimit = 100000000000ULL;
constexpr int kLineMax = 4 * (kIntegerDigits + 9) + 4;      // 4 numbers "i.ffffffff", 3 blanks, newline

struct Fixed8 {
  unsigned long long ip;  // integer part
  unsigned int frac;      // 8 fractional digits
  bool ok;                // false: outside the supported range (flag bit 0: mcgpu_write_formatted_projection fails with -3)
};

// round_half_even(v * 10^8) split at the decimal point; v >= 0 finite
__device__ __forceinline__ Fixed8 to_fixed8(double v) {
  Fixed8 r{0ULL, 0u, true};
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const int be = (int)((bits >> 52) & 0x7FF);
  unsigned long long m = bits & 0xFFFFFFFFFFFFFULL;
  int x;  // v = m * 2^x
  if (be == 0) x = -1074;
  else { m |= 1ULL << 52; x = be - 1075; }
  if (m == 0ULL) return r;
  if ((bits >> 63) != 0ULL || be == 0x7FF) { r.ok = false; return r; }
  // P = m * 10^8 as hi:lo
  const unsigned long long k = 100000000ULL;
  unsigned long long lo = m * k, hi = __umul64hi(m, k);
  unsigned long long R;  // the rounded product
  if (x >= 0) {
    if (hi != 0ULL || x >= 64 || (x > 0 && (lo >> (64 - x)) != 0ULL)) { r.ok = false; return r; }
    R = lo << x;
  } else {
    const int s = -x;
    if (s >= 128) return r;  // P < 2^80 <= half of 2^s: rounds to zero
    unsigned long long q, rem_hi, rem_lo, half_hi, half_lo;
    if (s >= 64) {
      q = (s == 64) ? hi : (hi >> (s - 64));
      rem_hi = (s == 64) ? 0ULL : (hi & ((1ULL << (s - 64)) - 1ULL));
      rem_lo = lo;
      half_hi = (s == 64) ? 0ULL : (1ULL << (s - 65));
      half_lo = (s == 64) ? (1ULL << 63) : 0ULL;
    } else {
      if ((hi >> s) != 0ULL) { r.ok = false; return r; }  // quotient beyond 64 bits
      q = (lo >> s) | (hi << (64 - s));
      rem_hi = 0ULL;
      rem_lo = lo & ((1ULL << s) - 1ULL);
      half_hi = 0ULL;
      half_lo = 1ULL << (s - 1);
    }
    const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
    const bool tie = rem_hi == half_hi && rem_lo == half_lo;
    R = q + ((above || (tie && (q & 1ULL))) ? 1ULL : 0ULL);
  }
  r.ip = R / k;
  r.frac = (unsigned int)(R - r.ip * k);
  if (r.ip >= kIntegerLimit) r.ok = false;  // more integer digits than the text buffers are sized for
  return r;
}

__device__ __forceinline__ int digits_of(unsigned long long v) {
  if ((v >> 32) == 0ULL) {
    const unsigned int u = (unsigned int)v;
    return 1 + (u >= 10u) + (u >= 100u) + (u >= 1000u) + (u >= 10000u) + (u >= 100000u) + (u >= 1000000u) + (u >= 10000000u) + (u >= 100000000u) +
           (u >= 1000000000u);
  }
  return 10 + (v >= 10000000000ULL);  // 2^32 <= v < 1e11
}

// the four numbers of pixel `pix` and the length of its line
__device__ __forceinline__ int format_pixel(const AsciiArgs& a, size_t pix, Fixed8 f[4], bool& ok, double& tot) {
  int n = 4 * 9 + 4;  // ".dddddddd" x 4 + separators
  tot = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double e = (double)a.image[pix + (size_t)c * a.npix];
    f[c] = to_fixed8(a.norm * e);
    ok = ok && f[c].ok;
    n += digits_of(f[c].ip);
    tot += e;
  }
  return n;
}

__device__ __forceinline__ unsigned char* put_number(unsigned char* w, const Fixed8& f) {
  const int nd = digits_of(f.ip);
  if ((f.ip >> 32) == 0ULL) {
    unsigned int v = (unsigned int)f.ip;
    for (int d = nd - 1; d >= 0; --d) { w[d] = (unsigned char)('0' + v % 10u); v /= 10u; }
  } else {
    unsigned long long v = f.ip;
    for (int d = nd - 1; d >= 0; --d) { w[d] = (unsigned char)('0' + (unsigned int)(v % 10ULL)); v /= 10ULL; }
  }
  w += nd;
  *w++ = '.';
  unsigned int q = f.frac;
#pragma unroll
  for (int d = 7; d >= 0; --d) { w[d] = (unsigned char)('0' + q % 10u); q /= 10u; }
  return w + 8;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(kThreads) void ascii_measure_kernel(AsciiArgs a) {
  __shared__ unsigned long long s_len[kWaves];
  __shared__ double s_sum[kWaves], s_max[kWaves];
  __shared__ long long s_arg[kWaves];
  __shared__ int s_bad;
  const int z = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_bad = 0;
  __syncthreads();
  unsigned long long len = 0;
  double sum = 0.0, mx = -100.0;
  long long arg = 0;
  bool ok = true;
  for (int x = t; x < a.nx; x += kThreads) {  // increasing pixel index: the strict ">" keeps the first of equal maxima
    const size_t pix = (size_t)z * a.nx + x;
    Fixed8 f[4];
    double tot;
    len += (unsigned long long)format_pixel(a, pix, f, ok, tot);
    sum += tot;  // integer-valued doubles: exact in any order
    if (tot > mx) { mx = tot; arg = (long long)pix; }
  }
  if (!ok) s_bad = 1;
  // largest value, then smallest pixel index (report_image's strict ">" in pixel order, MC-GPU_v1.3.cu:2893-2897)
  auto better = [](double m1, long long a1, double m2, long long a2) { return m1 > m2 || (m1 == m2 && a1 < a2); };
  len = wave_sum(len);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    const double m2 = __shfl_xor(mx, o);
    const long long a2 = __shfl_xor(arg, o);
    if (better(m2, a2, mx, arg)) { mx = m2; arg = a2; }
  }
  if (lane == 0) { s_len[wave] = len; s_sum[wave] = sum; s_max[wave] = mx; s_arg[wave] = arg; }
  __syncthreads();
  if (t == 0) {
    unsigned long long L = 1;  // the blank line after the row
    double S = 0.0, M = -100.0;
    long long A = 0;
    for (int i = 0; i < kWaves; ++i) {
      L += s_len[i];
      S += s_sum[i];
      if (i == 0 || better(s_max[i], s_arg[i], M, A)) { M = s_max[i]; A = s_arg[i]; }
    }
    a.row_len[z] = L;
    a.row_sum[z] = S;
    a.row_max[z] = M;
    a.row_arg[z] = A;
    if (s_bad) atomicOr(a.flags, 1u);
  }
}

__global__ __launch_bounds__(kThreads) void ascii_write_kernel(AsciiArgs a) {
  __shared__ unsigned long long s_part[kWaves];
  __shared__ unsigned int s_wave_len[kWaves];
  __shared__ __align__(16) unsigned char s_text[kThreads * kLineMax + 32];
  if (*a.flags & 1u) return;  // a value outside the formatter's range: no text; mcgpu_write_formatted_projection reports -3
  const int z = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // where this row starts: the lengths of the rows before it
  unsigned long long g = 0;
  for (int i = t; i < z; i += kThreads) g += a.row_len[i];
  g = wave_sum(g);
  if (lane == 0) s_part[wave] = g;
  __syncthreads();
  g = 0;
  for (int i = 0; i < kWaves; ++i) g += s_part[i];
  const unsigned long long row_len = a.row_len[z];
  if (t == 0) {
    a.row_off[z] = g;
    if (z == a.nz - 1) a.row_off[a.nz] = g + row_len;
  }
  if (g + row_len > a.capacity) {  // the same for every thread of the workgroup
    if (t == 0) atomicOr(a.flags, 2u);
    return;
  }
  for (int x0 = 0; x0 < a.nx; x0 += kThreads) {
    const int x = x0 + t;
    const bool act = x < a.nx;
    Fixed8 f[4];
    bool ok = true;
    double tot;
    const unsigned int len = act ? (unsigned int)format_pixel(a, (size_t)z * a.nx + x, f, ok, tot) : 0u;
    // exclusive scan of the line lengths over the tile
    unsigned int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    __syncthreads();  // the previous tile's text has been copied out; s_wave_len is free
    if (lane == 63) s_wave_len[wave] = inc;
    __syncthreads();
    unsigned int before = 0, tile_len = 0;
    for (int i = 0; i < kWaves; ++i) {
      if (i < wave) before += s_wave_len[i];
      tile_len += s_wave_len[i];
    }
    // the tile's text in LDS, at the phase of its file offset within 16 bytes: LDS and file are then aligned together
    const unsigned int phase = (unsigned int)(g & 15ULL);
    if (act) {
      unsigned char* w = s_text + phase + before + (inc - len);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        w = put_number(w, f[c]);
        *w++ = (c == 3) ? '\n' : ' ';
      }
    }
    __syncthreads();
    unsigned char* const dst = reinterpret_cast<unsigned char*>(a.text) + (g - phase);  // 16-byte aligned (hipMalloc'd base)
    const unsigned int end = phase + tile_len;
    for (unsigned int c = (unsigned int)t * 16u; c < end; c += kThreads * 16u) {
      if (c >= phase && c + 16u <= end) {
        *reinterpret_cast<uint4*>(dst + c) = *reinterpret_cast<const uint4*>(s_text + c);
      } else {
        const unsigned int lo = c > phase ? c : phase, hi = (c + 16u < end) ? c + 16u : end;
        for (unsigned int b = lo; b < hi; ++b) dst[b] = s_text[b];
      }
    }
    g += tile_len;
  }
  if (t == 0) a.text[g] = '\n';  // the blank line that closes the row
}

}  // namespace

hipError_t launch_ascii_format(const AsciiArgs& a, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(a.flags, 0, 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ascii_measure_kernel, dim3((unsigned)a.nz), dim3(kThreads), 0, stream, a);
  hipLaunchKernelGGL(ascii_write_kernel, dim3((unsigned)a.nz), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mcgpu
