// fdk.hip -- circular cone-beam FDK reconstruction for MI355X (what `rtkfdk --hardware cuda` does for
// the reference, cbctmc/reconstruction/reconstruction.py:22-69, reconstructors.py).  RTK itself is an un-vendored
// third-party dependency of the reference; the algorithm and RTK's geometry conventions are restated in oracle/fdk_oracle.py
// (parity unpinned against RTK, pinned by analytic phantoms) and implemented here as four kernels:
//   weight      : water pre-correction polynomial, cosine weight, displaced-detector (half-fan) weight, zero padding of the
//                 short side to a detector symmetric about the central ray                                     (streaming)
//   ramp        : rows zero-extended to L = 4096, batched hipFFT R2C -> multiply by the real spectrum of the (Hann-apodised)
//                 ramp -> C2R (HBM streaming); ramp_rows = the direct LDS convolution kept for A/B (MCGPU_FDK_DIRECT_RAMP)
//   extend_rows : --pad (RTK TruncationCorrection): every row continued on both sides by next = ceil(pad x width) columns with
//                 the feathered point reflection 2 p(border) - p(mirror), so that a truncated edge does not ring     (streaming)
//   smooth_cols : --hannY low-pass along v (3 taps for 1.0)                                                     (streaming)
//   backproject : voxel-driven, bilinear; a thread owns one (x, z) column of the volume, precomputes everything that does
//                 not depend on y for a batch of 8 projections in registers, then walks y: 4 loads + 10 flops per update;
//                 the filtered projections of a batch (5.7 MB each, padded) stay L2 / Infinity-Cache resident.  One voxel step
//                 is 3.9 detector pixels, so the 4 loads of a wave touch ~32 cache lines for 64 updates: the vector L1
//                 (64 B/clk/CU) bounds the kernel at ~1 update/clk/CU; measured 0.78                              (L1/L2 gather)
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcgpu_amd.h"
#include "knobs.hpp"

extern "C" void mcgpu_set_last_error_(const char* message);

namespace {

// projections per back-projection launch.  Measured at the reference's size (tools/fdk_bench.py): 8 projections and 4 waves per
// SIMD (78 VGPRs) 101 ms; 16 / 4 waves (128 VGPRs + scratch) 133 ms; 8 / 8 waves (scratch) 132 ms; 32 / 2 waves 215 ms
constexpr int kBatch = 8;

struct ProjParam {  // per projection, wave-uniform in the kernels
  float c, s;       // cos / sin of the gantry angle
  float off_x, off_y;
  float gap;        // angular weight [rad]: half the distance to both neighbouring projections (RTK GetAngularGaps)
};

// in: [n][nv][nu] raw line integrals; out: [n][nv][nu_p] weighted rows, padded with pad_l zero columns on the left (and zeros on the
// right) so that an off-centre detector becomes symmetric about the central ray (RTK: DisplacedDetectorImageFilter)
__global__ void weight_kernel(const float* __restrict__ in, float* __restrict__ out, int nu, int nv, int n, int nu_p /* row stride of out */, int pad_l, float du, float dv,
                              float u0, float v0, float sdd, const ProjParam* __restrict__ pp, const float* __restrict__ w_dis /*[n][nu]*/,
                              const float* __restrict__ wpc, int n_wpc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * nv * nu_p;
  if (i >= total) return;
  const int ip = (int)(i % nu_p), iv = (int)((i / nu_p) % nv), k = (int)(i / ((size_t)nu_p * nv));
  const int iu = ip - pad_l;
  float v = 0.f;
  if (iu >= 0 && iu < nu) {
    v = in[((size_t)k * nv + iv) * nu + iu];
    if (n_wpc > 0) {  // rtkfdk --wpc: sum_k c_k p^k
      float acc = 0.f, pw = 1.f;
      for (int j = 0; j < n_wpc; ++j) { acc += wpc[j] * pw; pw *= v; }
      v = acc;
    }
    const float up = u0 + du * iu + pp[k].off_x, vp = v0 + dv * iv + pp[k].off_y;
    v *= sdd / sqrtf(sdd * sdd + up * up + vp * vp) * w_dis[(size_t)k * nu + iu];
  }
  out[i] = v;
}

// rtkfdk --pad (rtk::FFTProjectionsConvolutionImageFilter::PadInputImageRegion with TruncationCorrection > 0; restated in
// oracle/fdk_oracle.py: truncation_extension).  A row occupies columns [next, next + n) of its buffer row; the columns at
// distance d = 1..next beyond either border get w[d] * (2 p(border) - p(border -/+ d)).  One thread per (row, d).
__global__ void extend_rows_kernel(float* __restrict__ rows, int stride, int n, int next, size_t n_rows, const float* __restrict__ w /*[next + 1]*/) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * (size_t)next) return;
  const int d = (int)(i % next) + 1;
  float* row = rows + (i / next) * (size_t)stride + next;
  row[-d] = w[d] * (2.0f * row[0] - row[d]);
  row[n - 1 + d] = w[d] * (2.0f * row[n - 1] - row[n - 1 - d]);
}

// out[row][i] = scale * sum_j in[row][j] * h[i - j + nu - 1];  one block per row, 256 threads, 4 consecutive outputs per thread
__global__ __launch_bounds__(256) void ramp_rows_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ h, int nu, float scale,
                                                        int j0, int j1 /* columns outside [j0, j1) are zero (padding) */) {
  extern __shared__ float lds[];
  float* row = lds;             // [nu]
  float* hk = lds + nu;         // [2 nu - 1 + 3] (padded with zeros so that the window may run past the ends)
  const size_t base = (size_t)blockIdx.x * nu;
  for (int i = threadIdx.x; i < nu; i += blockDim.x) row[i] = in[base + i];
  for (int i = threadIdx.x; i < 2 * nu + 2; i += blockDim.x) hk[i] = (i < 2 * nu - 1) ? h[i] : 0.f;
  __syncthreads();
  for (int i0 = 4 * threadIdx.x; i0 < nu; i0 += 4 * blockDim.x) {
    // window w_m = h[i0 + m - j + nu - 1], m = 0..3; stepping j -> j + 1 shifts it down by one
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int idx = i0 + nu - 1 - j0;  // index of w_0 for j = j0
    float w0 = hk[idx], w1 = hk[idx + 1], w2 = hk[idx + 2], w3 = hk[idx + 3];
    for (int j = j0; j < j1; ++j) {
      const float r = row[j];
      a0 = fmaf(r, w0, a0); a1 = fmaf(r, w1, a1); a2 = fmaf(r, w2, a2); a3 = fmaf(r, w3, a3);
      w3 = w2; w2 = w1; w1 = w0;
      --idx;
      w0 = (idx >= 0) ? hk[idx] : 0.f;
    }
    if (i0 + 0 < nu) out[base + i0 + 0] = a0 * scale;
    if (i0 + 1 < nu) out[base + i0 + 1] = a1 * scale;
    if (i0 + 2 < nu) out[base + i0 + 2] = a2 * scale;
    if (i0 + 3 < nu) out[base + i0 + 3] = a3 * scale;
  }
}

// in/out rows have `stride` floats, of which the first `nu` are used
__global__ void smooth_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int nu, int stride, int nv, int n, const float* __restrict__ ky, int nk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * nv * nu;
  if (i >= total) return;
  const int iu = (int)(i % nu), iv = (int)((i / nu) % nv), k = (int)(i / ((size_t)nu * nv));
  const size_t plane = (size_t)k * nv * stride;
  const int hk = nk / 2;
  float acc = 0.f;
  for (int j = 0; j < nk; ++j) {
    int r = iv + j - hk;
    r = r < 0 ? 0 : (r > nv - 1 ? nv - 1 : r);  // edge replicated
    acc += ky[j] * in[plane + (size_t)r * stride + iu];
  }
  out[plane + (size_t)iv * stride + iu] = acc;
}

// spectrum[row][k] *= H[k] (real: the ramp kernel is even), k = 0 .. L/2
__global__ void spectrum_kernel(float2* __restrict__ spec, const float* __restrict__ H, int nk, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float h = H[i % nk];
  float2 v = spec[i];
  v.x *= h; v.y *= h;
  spec[i] = v;
}

struct BackArgs {
  int nx, ny, nz, nu, nv, nb;  // nb = projections in this batch; nu = usable columns
  int u_first;                 // column of the buffer rows that holds detector column 0 (the --pad extension lies before it)
  int stride;                  // floats per detector row in q
  float x0, y0, z0, sx, sy, sz;
  float sid, sdd, inv_du, inv_dv, u0, v0;
  ProjParam pp[kBatch];
};

__global__ __launch_bounds__(256, 4) void backproject_kernel(float* __restrict__ vol, const float* __restrict__ q /*[nb][nv][nu]*/, const BackArgs A) {
  const int ix = blockIdx.x * blockDim.x + threadIdx.x, iz = blockIdx.y;
  if (ix >= A.nx) return;
  const float X = A.x0 + A.sx * ix, Z = A.z0 + A.sz * iz;
  int iu[kBatch];
  float au[kBatch], wg[kBatch], av_a[kBatch], av_b[kBatch];
#pragma unroll
  for (int k = 0; k < kBatch; ++k) {
    iu[k] = -1; au[k] = 0.f; wg[k] = 0.f; av_a[k] = 0.f; av_b[k] = 0.f;
    if (k < A.nb) {
      const float xr = X * A.pp[k].c - Z * A.pp[k].s, zr = X * A.pp[k].s + Z * A.pp[k].c;
      const float U = A.sid - zr, mag = A.sdd / U;
      const float fu = (mag * xr - A.pp[k].off_x - A.u0) * A.inv_du;
      const float fl = floorf(fu);
      const int i = (int)fl;
      if (i >= 0 && i < A.nu - 1) {
        iu[k] = i;
        au[k] = fu - fl;
        const float r = A.sid / U;
        wg[k] = A.pp[k].gap * r * r;
        av_a[k] = mag * A.inv_dv;                                  // fv = av_a * Y + av_b
        av_b[k] = (-A.pp[k].off_y - A.v0) * A.inv_dv;
      }
    }
  }
  const size_t plane = (size_t)A.stride * A.nv;
  float* out = vol + ((size_t)iz * A.ny) * A.nx + ix;
  for (int iy = 0; iy < A.ny; ++iy) {
    const float Y = A.y0 + A.sy * iy;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (iu[k] >= 0) {
        const float fv = fmaf(av_a[k], Y, av_b[k]);
        const float fl = floorf(fv);
        const int iv = (int)fl;
        if (iv >= 0 && iv < A.nv - 1) {
          const float av = fv - fl;
          const float* r0 = q + (size_t)k * plane + (size_t)iv * A.stride + A.u_first + iu[k];
          float2 lo, hi;  // (iu, iu + 1) of both rows with one 8-byte load each (4-byte aligned: global loads need no more)
          __builtin_memcpy(&lo, r0, 8);
          __builtin_memcpy(&hi, r0 + A.stride, 8);
          const float v00 = lo.x, v01 = lo.y, v10 = hi.x, v11 = hi.y;
          const float top = fmaf(au[k], v01 - v00, v00), bot = fmaf(au[k], v11 - v10, v10);
          acc = fmaf(wg[k], fmaf(av, bot - top, top), acc);
        }
      }
    }
    out[(size_t)iy * A.nx] += acc;
  }
}

// ---- host helpers ------------------------------------------------------------------------------------------------
void fft(std::vector<std::complex<double>>& a, bool inverse) {  // radix 2, in place
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = 2.0 * M_PI / (double)len * (inverse ? 1.0 : -1.0);
    const std::complex<double> wl(std::cos(ang), std::sin(ang));
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w(1.0, 0.0);
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
        w *= wl;
      }
    }
  }
  if (inverse)
    for (auto& x : a) x /= (double)n;
}

// oracle/fdk_oracle.py: ramp_kernel
std::vector<double> ramp_kernel(int n_half, double hann) {
  std::vector<double> h(2 * n_half + 1, 0.0);
  for (int n = -n_half; n <= n_half; ++n) {
    if (n == 0) h[n + n_half] = 0.25;
    else if (n % 2 != 0) h[n + n_half] = -1.0 / (M_PI * M_PI * (double)n * (double)n);
  }
  if (hann > 0.0) {
    size_t m = 1;
    while (m < (size_t)8 * (2 * n_half + 1)) m *= 2;
    std::vector<std::complex<double>> buf(m, 0.0);
    for (int i = 0; i <= n_half; ++i) buf[i] = h[n_half + i];
    for (int i = 1; i <= n_half; ++i) buf[m - i] = h[n_half - i];
    fft(buf, false);
    const double fc = 0.5 * hann;
    for (size_t i = 0; i < m; ++i) {
      const double f = (i < m / 2) ? (double)i / (double)m : (double)i / (double)m - 1.0;
      const double win = (std::fabs(f) < fc) ? 0.5 * (1.0 + std::cos(M_PI * f / fc)) : 0.0;
      buf[i] *= win;
    }
    fft(buf, true);
    for (int i = 0; i <= n_half; ++i) h[n_half + i] = buf[i].real();
    for (int i = 1; i <= n_half; ++i) h[n_half - i] = buf[m - i].real();
  }
  return h;
}

std::vector<double> hann_y_kernel(double hann_y) {
  if (hann_y <= 0.0) return {1.0};
  if (hann_y == 1.0) return {0.25, 0.5, 0.25};
  const size_t m = 4096;
  const int n_half = 8;
  std::vector<std::complex<double>> buf(m);
  const double fc = 0.5 * hann_y;
  for (size_t i = 0; i < m; ++i) {
    const double f = (i < m / 2) ? (double)i / (double)m : (double)i / (double)m - 1.0;
    buf[i] = (std::fabs(f) < fc) ? 0.5 * (1.0 + std::cos(M_PI * f / fc)) : 0.0;
  }
  fft(buf, true);
  std::vector<double> k(2 * n_half + 1);
  for (int i = 0; i <= n_half; ++i) k[n_half + i] = buf[i].real();
  for (int i = 1; i <= n_half; ++i) k[n_half - i] = buf[m - i].real();
  double sum = 0.0;
  for (double v : k) sum += v;
  for (double& v : k) v /= sum;  // truncated support: keep the DC gain at exactly 1
  return k;
}

// oracle/fdk_oracle.py: displaced_weights (Wang 2002) for the columns of one projection
void displaced_weights(int nu, double du, double u0, double off_x, double sdd, float* w) {
  const double lo = u0 + off_x, hi = u0 + du * (nu - 1) + off_x;
  if (lo >= 0.0 || hi <= 0.0) { for (int i = 0; i < nu; ++i) w[i] = 1.f; return; }
  const double theta = std::min(-lo, hi);
  if (std::fabs((-lo) - hi) < 1e-9 * std::max(-lo, hi)) { for (int i = 0; i < nu; ++i) w[i] = 0.5f; return; }
  const double sign = (hi > -lo) ? 1.0 : -1.0;
  for (int i = 0; i < nu; ++i) {
    const double s = sign * (u0 + du * i + off_x);
    double v = (s > theta) ? 1.0 : 0.0;
    if (std::fabs(s) <= theta) v = 0.5 * (std::sin(M_PI * std::atan(s / sdd) / (2.0 * std::atan(theta / sdd))) + 1.0);
    w[i] = (float)v;
  }
}

struct FdkError { std::string msg; };
#define FDK_HIP(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess) throw FdkError{std::string("!!HIP ERROR!! ") + #expr + ": " + hipGetErrorString(_e)}; \
  } while (0)

}  // namespace

extern "C" int mcgpu_fdk_reconstruct(const mcgpu_fdk_options* caller_o, const float* projections, float* volume, mcgpu_fdk_report* report) {
  if (!caller_o || caller_o->struct_size < 2 * sizeof(int)) {
    mcgpu_set_last_error_("!!ERROR!! mcgpu_fdk_reconstruct: set mcgpu_fdk_options.struct_size = sizeof(mcgpu_fdk_options)");
    return -1;
  }
  // a caller built against an older header passes a shorter struct: what it does not have (e.g. `pad`) reads as zero
  mcgpu_fdk_options local;
  memset(&local, 0, sizeof local);
  memcpy(&local, caller_o, std::min<size_t>(caller_o->struct_size, sizeof local));
  const mcgpu_fdk_options* o = &local;
  if (!projections || !volume || o->n_proj < 1 || o->nu < 2 || o->nv < 2 || o->nx < 1 || o->ny < 1 || o->nz < 1 || !o->gantry_deg ||
      !(o->du > 0) || !(o->dv > 0) || !(o->sid > 0) || !(o->sdd > 0)) {
    mcgpu_set_last_error_("!!ERROR!! mcgpu_fdk_reconstruct: bad argument");
    return -1;
  }
  float2* d_spec = nullptr;
  hipfftHandle plan_fwd = 0, plan_inv = 0;
  float *d_free_raw = nullptr, *d_free_wext = nullptr, *d_in = nullptr, *d_tmp = nullptr, *d_vol = nullptr, *d_h = nullptr, *d_ky = nullptr, *d_wdis = nullptr, *d_wpc = nullptr;
  ProjParam* d_pp = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
  int rc = 0;
  try {
    FDK_HIP(hipSetDevice(o->device));
    const int n = o->n_proj, nu = o->nu, nv = o->nv;
    const size_t plane = (size_t)nu * nv, nvox = (size_t)o->nx * o->ny * o->nz;
    const std::vector<double> kyd = hann_y_kernel(o->hann_y);
    std::vector<float> ky(kyd.begin(), kyd.end());
    std::vector<ProjParam> pp(n);
    std::vector<float> wdis((size_t)n * nu);
    for (int k = 0; k < n; ++k) {
      const double t = o->gantry_deg[k] * M_PI / 180.0;
      const double ox = o->proj_offset_x ? o->proj_offset_x[k] : 0.0, oy = o->proj_offset_y ? o->proj_offset_y[k] : 0.0;
      pp[k] = {(float)std::cos(t), (float)std::sin(t), (float)ox, (float)oy, 0.f};
      displaced_weights(nu, o->du, o->u0, ox, o->sdd, &wdis[(size_t)k * nu]);
    }
    {
      // Angular weight of a projection = half the distance to its two neighbours on the circle (what rtkfdk takes from
      // the geometry file: ThreeDCircularProjectionGeometry::GetAngularGaps); 2 pi / n only for a uniform full arc.
      // Projections at the same angle share their gap.
      std::vector<std::pair<double, int>> by_angle(n);
      for (int k = 0; k < n; ++k) {
        double a = std::fmod(o->gantry_deg[k], 360.0);
        if (a < 0) a += 360.0;
        by_angle[k] = {a, k};
      }
      std::sort(by_angle.begin(), by_angle.end());
      std::vector<double> uniq;
      std::vector<int> count;
      for (int k = 0; k < n; ++k) {
        if (uniq.empty() || by_angle[k].first - uniq.back() > 1e-9) { uniq.push_back(by_angle[k].first); count.push_back(1); }
        else ++count.back();
      }
      const int m = (int)uniq.size();
      int u = -1;
      double last = -1.0;
      for (int k = 0; k < n; ++k) {
        if (u < 0 || by_angle[k].first - last > 1e-9) { ++u; last = uniq[u]; }
        double gap = 360.0;
        if (m > 1) {
          const double prev = uniq[(u + m - 1) % m], next = uniq[(u + 1) % m];
          double d = next - prev;
          if (d <= 0) d += 360.0;
          gap = 0.5 * d;
          if (m == 2) gap = 180.0;
        }
        pp[by_angle[k].second].gap = (float)(gap * M_PI / 180.0 / count[u]);
      }
    }
    const double ox0 = std::isnan(o->ox) ? -(o->nx - 1) / 2.0 * o->sx : o->ox, oy0 = std::isnan(o->oy) ? -(o->ny - 1) / 2.0 * o->sy : o->oy,
                 oz0 = std::isnan(o->oz) ? -(o->nz - 1) / 2.0 * o->sz : o->oz;
    // symmetric padding of an off-centre detector (oracle/fdk_oracle.py: symmetric_padding)
    int pad_l = 0, pad_r = 0;
    {
      double off_min = 1e300, off_max = -1e300;
      for (int k = 0; k < n; ++k) { off_min = std::min(off_min, (double)pp[k].off_x); off_max = std::max(off_max, (double)pp[k].off_x); }
      const double last = o->u0 + (nu - 1) * o->du;
      const double lo = o->u0 + off_min, hi = last + off_max;
      if (lo < 0.0 && hi > 0.0) {
        const double extent = std::max(std::max(-(o->u0 + off_min), -(o->u0 + off_max)), std::max(last + off_min, last + off_max));
        pad_l = std::max(0, (int)std::ceil((extent + (o->u0 + off_min)) / o->du - 1e-9));
        pad_r = std::max(0, (int)std::ceil((extent - (last + off_max)) / o->du - 1e-9));
      }
    }
    const int nu_p = nu + pad_l + pad_r;
    const double u0_p = o->u0 - pad_l * o->du;
    // rtkfdk --pad: the ramp sees rows of nu_e = nu_p + 2 next columns (extend_rows_kernel); the back-projector only the nu_p
    // detector columns in their middle
    const int next = (o->pad > 0.0) ? std::min((int)std::ceil(o->pad * nu_p), nu_p - 1) : 0;
    const int nu_e = nu_p + 2 * next;
    // Ramp filter: FFT (hipFFT, rows zero-extended to L >= 2 nu_p - 1: no wrap-around inside the nu_p columns that are used) or,
    // with MCGPU_FDK_DIRECT_RAMP, the direct LDS convolution (same result up to float rounding; tests compare both to the oracle)
    const bool direct = mcgpu::knob_set("MCGPU_FDK_DIRECT_RAMP");
    // The ramp is a linear convolution evaluated as a circular one of length L.  Only the nu_p detector columns in the middle of
    // a row are ever read, and for those the lag between an output and any of the nu_e data columns is at most M = nu_p + next - 1:
    // with the kernel cut to |lag| <= M, L >= 2 M + 1 keeps every lag distinct (and L >= nu_e holds the row).  L = the smallest
    // even 2^a 3^b 5^c at or above that -- 7680 instead of 16384 for the reference's half-fan rows with pad = 1.
    const int max_lag = nu_p + next - 1;
    int L = std::max(2 * max_lag + 1, nu_e);
    for (;; ++L) {
      if (L & 1) continue;
      int m = L;
      for (int f : {2, 3, 5})
        while (m % f == 0) m /= f;
      if (m == 1) break;
    }
    const int stride = direct ? nu_e : L;        // floats per detector row in the filtered buffers
    const int nk = L / 2 + 1;
    const size_t plane_p = (size_t)stride * nv;
    const int chunk = std::min(n, direct ? 64 : 32);  // projections resident on the device at a time (multiple of kBatch)
    float* d_raw = nullptr;
    FDK_HIP(hipMalloc(&d_raw, (size_t)chunk * plane * 4));
    d_free_raw = d_raw;
    FDK_HIP(hipMalloc(&d_in, (size_t)chunk * plane_p * 4));
    FDK_HIP(hipMalloc(&d_tmp, (size_t)chunk * plane_p * 4));
    FDK_HIP(hipMalloc(&d_vol, nvox * 4));
    FDK_HIP(hipMemset(d_vol, 0, nvox * 4));
    const std::vector<double> hd = ramp_kernel(nu_e - 1, o->hann);
    float* d_wext = nullptr;
    if (next > 0) {
      std::vector<float> wext((size_t)next + 1, 0.f);
      for (int d = 1; d <= next; ++d) wext[(size_t)d] = next > 1 ? (float)std::pow(std::sin((double)(next - d) * M_PI / (2.0 * next - 2.0)), 0.75) : 0.f;
      FDK_HIP(hipMalloc(&d_wext, wext.size() * 4));
      d_free_wext = d_wext;
      FDK_HIP(hipMemcpy(d_wext, wext.data(), wext.size() * 4, hipMemcpyHostToDevice));
    }
    const double scale = (o->sdd / o->sid) / o->du;
    if (direct) {
      std::vector<float> h(hd.begin(), hd.end());
      FDK_HIP(hipMalloc(&d_h, h.size() * 4));
      FDK_HIP(hipMemcpy(d_h, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    } else {
      // spectrum of the kernel laid out circularly (lag n at index n mod L); real because the kernel is even;
      // the scale of the filter and hipFFT's missing 1/L are folded in
      // L is not a power of two: the (real, even) kernel's spectrum by its cosine sum, H[k] = h[0] + 2 sum_lag h[lag] cos(2 pi k lag / L),
      // with one table of cosines (30 M multiply-adds in double: tens of milliseconds, once per reconstruction)
      std::vector<double> cosine((size_t)L);
      for (int t = 0; t < L; ++t) cosine[(size_t)t] = std::cos(2.0 * M_PI * (double)t / (double)L);
      std::vector<float> H((size_t)nk);
      const double* h0 = hd.data() + (nu_e - 1);  // h0[lag], lag = -(nu_e - 1) .. nu_e - 1
      for (int k = 0; k < nk; ++k) {
        double acc = h0[0];
        size_t t = 0;  // (k * lag) mod L
        for (int lag = 1; lag <= max_lag; ++lag) {
          t += (size_t)k;
          if (t >= (size_t)L) t -= (size_t)L;
          acc += 2.0 * h0[lag] * cosine[t];
        }
        H[(size_t)k] = (float)(acc * scale / (double)L);
      }
      FDK_HIP(hipMalloc(&d_h, H.size() * 4));
      FDK_HIP(hipMemcpy(d_h, H.data(), H.size() * 4, hipMemcpyHostToDevice));
      FDK_HIP(hipMalloc(&d_spec, (size_t)chunk * nv * nk * sizeof(float2)));
    }
    FDK_HIP(hipMalloc(&d_ky, ky.size() * 4));
    FDK_HIP(hipMemcpy(d_ky, ky.data(), ky.size() * 4, hipMemcpyHostToDevice));
    FDK_HIP(hipMalloc(&d_wdis, wdis.size() * 4));
    FDK_HIP(hipMemcpy(d_wdis, wdis.data(), wdis.size() * 4, hipMemcpyHostToDevice));
    FDK_HIP(hipMalloc(&d_pp, pp.size() * sizeof(ProjParam)));
    FDK_HIP(hipMemcpy(d_pp, pp.data(), pp.size() * sizeof(ProjParam), hipMemcpyHostToDevice));
    std::vector<float> wpc;
    for (int j = 0; j < o->n_wpc; ++j) wpc.push_back((float)o->wpc[j]);
    if (!wpc.empty()) {
      FDK_HIP(hipMalloc(&d_wpc, wpc.size() * 4));
      FDK_HIP(hipMemcpy(d_wpc, wpc.data(), wpc.size() * 4, hipMemcpyHostToDevice));
    }
    FDK_HIP(hipEventCreate(&e0)); FDK_HIP(hipEventCreate(&e1)); FDK_HIP(hipEventCreate(&e2)); FDK_HIP(hipEventCreate(&e3));
    double ms_filter = 0.0, ms_back = 0.0;
    const size_t lds_ramp = ((size_t)nu_e + 2 * nu_e + 2) * 4;
    if (direct && lds_ramp > 64 * 1024)
      FDK_HIP(hipFuncSetAttribute((const void*)ramp_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ramp));
    int planned_rows = 0;
    for (int first = 0; first < n; first += chunk) {
      const int m = std::min(chunk, n - first);
      const size_t elems = (size_t)m * plane_p;
      FDK_HIP(hipMemcpy(d_raw, projections + (size_t)first * plane, (size_t)m * plane * 4, hipMemcpyHostToDevice));
      if (!direct && planned_rows != m * nv) {  // one batched plan per chunk size (at most two: full chunks and the last one)
        if (plan_fwd) { hipfftDestroy(plan_fwd); hipfftDestroy(plan_inv); plan_fwd = plan_inv = 0; }
        int len[1] = {L};
        if (hipfftPlanMany(&plan_fwd, 1, len, nullptr, 1, L, nullptr, 1, nk, HIPFFT_R2C, m * nv) != HIPFFT_SUCCESS ||
            hipfftPlanMany(&plan_inv, 1, len, nullptr, 1, nk, nullptr, 1, L, HIPFFT_C2R, m * nv) != HIPFFT_SUCCESS)
          throw FdkError{"!!ERROR!! mcgpu_fdk_reconstruct: hipfftPlanMany failed"};
        planned_rows = m * nv;
      }
      FDK_HIP(hipEventRecord(e0, nullptr));
      const unsigned gb = (unsigned)((elems + 255) / 256);
      hipLaunchKernelGGL(weight_kernel, dim3(gb), dim3(256), 0, nullptr, d_raw, d_in, nu, nv, m, stride, next + pad_l, (float)o->du, (float)o->dv, (float)o->u0,
                         (float)o->v0, (float)o->sdd, d_pp + first, d_wdis + (size_t)first * nu, d_wpc, (int)wpc.size());
      if (next > 0) {
        const size_t ne = (size_t)m * nv * next;
        hipLaunchKernelGGL(extend_rows_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, nullptr, d_in, stride, nu_p, next, (size_t)m * nv, d_wext);
      }
      if (direct) {
        hipLaunchKernelGGL(ramp_rows_kernel, dim3((unsigned)(m * nv)), dim3(256), lds_ramp, nullptr, d_in, d_tmp, d_h, nu_e, (float)scale, next > 0 ? 0 : pad_l,
                           next > 0 ? nu_e : pad_l + nu);
      } else {
        if (hipfftExecR2C(plan_fwd, d_in, (hipfftComplex*)d_spec) != HIPFFT_SUCCESS) throw FdkError{"!!ERROR!! mcgpu_fdk_reconstruct: hipfftExecR2C failed"};
        const size_t ns = (size_t)m * nv * nk;
        hipLaunchKernelGGL(spectrum_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nullptr, d_spec, d_h, nk, ns);
        if (hipfftExecC2R(plan_inv, (hipfftComplex*)d_spec, d_tmp) != HIPFFT_SUCCESS) throw FdkError{"!!ERROR!! mcgpu_fdk_reconstruct: hipfftExecC2R failed"};
      }
      const float* filtered = d_tmp;
      if (ky.size() > 1) {
        const size_t na = (size_t)m * nv * nu_e;
        hipLaunchKernelGGL(smooth_cols_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, nullptr, d_tmp, d_in, nu_e, stride, nv, m, d_ky, (int)ky.size());
        filtered = d_in;
      }
      FDK_HIP(hipEventRecord(e1, nullptr));
      for (int b = 0; b < m; b += kBatch) {
        BackArgs A;
        A.nx = o->nx; A.ny = o->ny; A.nz = o->nz; A.nu = nu_p; A.u_first = next; A.stride = stride; A.nv = nv; A.nb = std::min(kBatch, m - b);
        A.x0 = (float)ox0; A.y0 = (float)oy0; A.z0 = (float)oz0; A.sx = (float)o->sx; A.sy = (float)o->sy; A.sz = (float)o->sz;
        A.sid = (float)o->sid; A.sdd = (float)o->sdd; A.inv_du = (float)(1.0 / o->du); A.inv_dv = (float)(1.0 / o->dv);
        A.u0 = (float)u0_p; A.v0 = (float)o->v0;
        for (int k = 0; k < kBatch; ++k) A.pp[k] = (k < A.nb) ? pp[first + b + k] : ProjParam{1.f, 0.f, 0.f, 0.f, 0.f};
        hipLaunchKernelGGL(backproject_kernel, dim3((unsigned)((o->nx + 255) / 256), (unsigned)o->nz), dim3(256), 0, nullptr, d_vol,
                           filtered + (size_t)b * plane_p, A);
      }
      FDK_HIP(hipEventRecord(e2, nullptr));
      FDK_HIP(hipEventSynchronize(e2));
      float a = 0.f, bms = 0.f;
      FDK_HIP(hipEventElapsedTime(&a, e0, e1));
      FDK_HIP(hipEventElapsedTime(&bms, e1, e2));
      ms_filter += a; ms_back += bms;
    }
    FDK_HIP(hipGetLastError());
    FDK_HIP(hipMemcpy(volume, d_vol, nvox * 4, hipMemcpyDeviceToHost));
    if (report) { report->ms_filter = ms_filter; report->ms_backproject = ms_back; }
  } catch (const FdkError& e) {
    mcgpu_set_last_error_(e.msg.c_str());
    rc = -1;
  }
  if (plan_fwd) hipfftDestroy(plan_fwd);
  if (plan_inv) hipfftDestroy(plan_inv);
  for (void* p : {(void*)d_spec, (void*)d_free_raw, (void*)d_in, (void*)d_tmp, (void*)d_vol, (void*)d_h, (void*)d_ky, (void*)d_wdis, (void*)d_wpc, (void*)d_pp, (void*)d_free_wext})
    if (p) (void)hipFree(p);
  for (hipEvent_t e : {e0, e1, e2, e3})
    if (e) (void)hipEventDestroy(e);
  return rc;
}
