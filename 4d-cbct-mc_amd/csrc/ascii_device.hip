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
// Two passes over the tallies, one workgroup per detector row (the lines of a row are contiguous in the file):
//   ascii_measure_kernel  line lengths -> bytes per row; row sums and the row's largest pixel for the footer
//   ascii_offsets_kernel  exclusive scan over the rows (768 values, one workgroup)
//   ascii_write_kernel    every thread re-formats its run of pixels and writes it at its offset
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ascii_device.hpp"

namespace mcgpu {
namespace {

constexpr int kThreads = 256;

struct Fixed8 {
  unsigned long long ip;  // integer part
  unsigned int frac;      // 8 fractional digits
  bool ok;                // false: outside the supported range (the host formats that projection)
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
  r.frac = (unsigned int)(R % k);
  return r;
}

__device__ __forceinline__ int digits_of(unsigned long long v) {
  int n = 1;
  while (v >= 10ULL) { v /= 10ULL; ++n; }
  return n;
}

// characters of one pixel line: 4 numbers, 3 blanks, newline
__device__ __forceinline__ int line_length(const AsciiArgs& a, size_t pix, bool& ok, double& tot) {
  int n = 4 * 9 + 4;  // ".dddddddd" x 4 + separators
  tot = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double e = (double)a.image[pix + (size_t)c * a.npix];
    const Fixed8 f = to_fixed8(a.norm * e);
    ok = ok && f.ok;
    n += digits_of(f.ip);
    tot += e;
  }
  return n;
}

__device__ __forceinline__ char* put_number(char* w, const Fixed8& f) {
  char tmp[20];
  int n = 0;
  unsigned long long v = f.ip;
  do { tmp[n++] = (char)('0' + (int)(v % 10ULL)); v /= 10ULL; } while (v != 0ULL);
  while (n) *w++ = tmp[--n];
  *w++ = '.';
  unsigned int q = f.frac;
#pragma unroll
  for (int d = 7; d >= 0; --d) { w[d] = (char)('0' + (int)(q % 10u)); q /= 10u; }
  return w + 8;
}

__global__ __launch_bounds__(kThreads) void ascii_measure_kernel(AsciiArgs a) {
  __shared__ unsigned long long s_len[kThreads];
  __shared__ double s_sum[kThreads], s_max[kThreads];
  __shared__ long long s_arg[kThreads];
  __shared__ int s_bad;
  const int z = blockIdx.x, t = threadIdx.x;
  if (t == 0) s_bad = 0;
  __syncthreads();
  const int chunk = (a.nx + kThreads - 1) / kThreads, x0 = t * chunk, x1 = min(x0 + chunk, a.nx);
  unsigned long long len = 0;
  double sum = 0.0, mx = -100.0;
  long long arg = 0;
  bool ok = true;
  for (int x = x0; x < x1; ++x) {
    const size_t pix = (size_t)z * a.nx + x;
    double tot;
    len += (unsigned long long)line_length(a, pix, ok, tot);
    sum += tot;
    if (tot > mx) { mx = tot; arg = (long long)pix; }
  }
  if (!ok) s_bad = 1;
  s_len[t] = len; s_sum[t] = sum; s_max[t] = mx; s_arg[t] = arg;
  __syncthreads();
  if (t == 0) {  // in thread order: the first of equal maxima wins, like report_image's strict ">"
    unsigned long long L = 1;  // the blank line after the row
    double S = 0.0, M = -100.0;
    long long A = 0;
    for (int i = 0; i < kThreads; ++i) {
      L += s_len[i];
      S += s_sum[i];
      if (s_max[i] > M) { M = s_max[i]; A = s_arg[i]; }
    }
    a.row_len[z] = L;
    a.row_sum[z] = S;
    a.row_max[z] = M;
    a.row_arg[z] = A;
    if (s_bad) atomicOr(a.flags, 1u);
  }
}

__global__ __launch_bounds__(kThreads) void ascii_offsets_kernel(AsciiArgs a) {
  if (threadIdx.x == 0) {
    unsigned long long off = 0;
    for (int z = 0; z < a.nz; ++z) { a.row_off[z] = off; off += a.row_len[z]; }
    a.row_off[a.nz] = off;
    if (off > a.capacity) atomicOr(a.flags, 2u);
  }
}

__global__ __launch_bounds__(kThreads) void ascii_write_kernel(AsciiArgs a) {
  __shared__ unsigned long long s_off[kThreads];
  if (*a.flags != 0u) return;  // unsupported value or too small a buffer: the host formats this projection
  const int z = blockIdx.x, t = threadIdx.x;
  const int chunk = (a.nx + kThreads - 1) / kThreads, x0 = t * chunk, x1 = min(x0 + chunk, a.nx);
  unsigned long long len = 0;
  bool ok = true;
  for (int x = x0; x < x1; ++x) {
    double tot;
    len += (unsigned long long)line_length(a, (size_t)z * a.nx + x, ok, tot);
  }
  s_off[t] = len;
  __syncthreads();
  if (t == 0) {
    unsigned long long off = a.row_off[z];
    for (int i = 0; i < kThreads; ++i) { const unsigned long long l = s_off[i]; s_off[i] = off; off += l; }
  }
  __syncthreads();
  char* w = a.text + s_off[t];
  for (int x = x0; x < x1; ++x) {
    const size_t pix = (size_t)z * a.nx + x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w = put_number(w, to_fixed8(a.norm * (double)a.image[pix + (size_t)c * a.npix]));
      *w++ = (c == 3) ? '\n' : ' ';
    }
  }
  if (x1 == a.nx && x0 < a.nx) *w = '\n';  // the thread that holds the row's last pixel closes the row
}

}  // namespace

hipError_t launch_ascii_format(const AsciiArgs& a, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(a.flags, 0, 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ascii_measure_kernel, dim3((unsigned)a.nz), dim3(kThreads), 0, stream, a);
  hipLaunchKernelGGL(ascii_offsets_kernel, dim3(1), dim3(kThreads), 0, stream, a);
  hipLaunchKernelGGL(ascii_write_kernel, dim3((unsigned)a.nz), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mcgpu
