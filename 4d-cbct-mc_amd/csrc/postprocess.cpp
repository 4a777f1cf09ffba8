// postprocess.cpp -- projection post-processing and MetaImage stacks.
//
// The reference turns the per-projection ASCII files into RTK-ready stacks in Python
// (cbctmc/mc/projection.py:36-169, cbctmc/mc/simulation.py:235-277): np.loadtxt -> float32 -> reshape (Nz, Nx, 4)
// -> flip z -> crop the half-fan columns -> per mode {total, unscattered, scattered} sum the classes -> replace zeros
// by the smallest positive value of the stack -> SimpleITK image (spacing = pixel size, origin = -size/2) -> .mha;
// and, with an air scan, projections_total_normalized = log(gaussian_filter(air, sigma) / total).
// Here the same numbers are produced straight from the integer tallies (decimal8.hpp), streamed plane by plane
// into MetaImage files, without the 63 MB-per-projection text detour.
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "decimal8.hpp"
#include "host_model.hpp"

namespace mcgpu {

// ------------------------------------------------------------------------------------------------
// u64 tallies [4][Nz][Nx] -> float32 planes [3][Nz][crop_nx] (total, unscattered, scattered), z flipped
// ------------------------------------------------------------------------------------------------
void finalize_projection_host(const HostModel& m, const uint64_t* image, unsigned long long total_histories, int crop_nx, float* planes,
                              int n_threads) {
  const DetectorPose& d0 = m.detector[0];
  const int nx = d0.nx, nz = d0.nz;
  const int cx = (crop_nx > 0 && crop_nx < nx) ? crop_nx : nx;
  const size_t npix = (size_t)nx * nz, plane = (size_t)cx * nz;
  const double SCALE = 1.0 / 100.0f;
  const double norm = SCALE * d0.inv_pixel_size_X * d0.inv_pixel_size_Z / ((double)total_histories);
  int T = n_threads > 0 ? n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  T = std::max(1, std::min(T, nz));
  auto work = [&](int t) {
    for (int zo = (int)((long)nz * t / T); zo < (int)((long)nz * (t + 1) / T); ++zo) {
      const size_t src = (size_t)(nz - 1 - zo) * nx, dst = (size_t)zo * cx;
      for (int x = 0; x < cx; ++x)
        finalize_pixel(image[src + x], image[src + x + npix], image[src + x + 2 * npix], image[src + x + 3 * npix], norm, planes[dst + x],
                       planes[plane + dst + x], planes[2 * plane + dst + x]);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

// ------------------------------------------------------------------------------------------------
// MetaImage (.mha) float32 stack, written plane by plane.  Header fields as ITK's MetaImageIO writes them for
// sitk.GetImageFromArray(stack) with SetSpacing((sx, sy, 1)) and SetOrigin((-nx*sx/2, -ny*sy/2, 0))
// (projection.py:155-164).
// ------------------------------------------------------------------------------------------------
struct MhaStack {
  FILE* fp = nullptr;
  std::string path;
  int nx = 0, ny = 0, nslices = 0, written = 0;
  long data_offset = 0;
  float min_positive = std::numeric_limits<float>::infinity();
  std::vector<uint64_t> zeros;  // element indices of the exact zeros of slices that hold only a few (patched in place by finish)
  std::vector<uint32_t> zero_count;  // exact zeros per slice (a slice with many is rewritten whole by finish)
  std::vector<unsigned char> have;  // slices written so far (random-access writes of a 4-D scan)
  std::mutex mu;                    // writes by slice index may come from several scans at once (projection-sharded devices)
};

static std::string fmt_g(double v) {
  char b[64];
  snprintf(b, sizeof b, "%.17g", v);
  double back = strtod(b, nullptr);
  for (int p = 1; p < 17; ++p) {  // shortest representation that round-trips
    char s[64];
    snprintf(s, sizeof s, "%.*g", p, v);
    if (strtod(s, nullptr) == back) return s;
  }
  return b;
}

MhaStack* mha_create(const std::string& path, int nx, int ny, int nslices, double sx, double sy) {
  FILE* fp = fopen(path.c_str(), "wb+");
  if (!fp) throw Error(-3, "!!ERROR!! can not open " + path + " for writing");
  MhaStack* s = new MhaStack;
  s->fp = fp; s->path = path; s->nx = nx; s->ny = ny; s->nslices = nslices;
  std::string h;
  h += "ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = False\n";
  h += "TransformMatrix = 1 0 0 0 1 0 0 0 1\n";
  h += "Offset = " + fmt_g(-nx * sx / 2) + " " + fmt_g(-ny * sy / 2) + " 0\n";
  h += "CenterOfRotation = 0 0 0\nAnatomicalOrientation = RAI\n";
  h += "ElementSpacing = " + fmt_g(sx) + " " + fmt_g(sy) + " 1\n";
  h += "DimSize = " + std::to_string(nx) + " " + std::to_string(ny) + " " + std::to_string(nslices) + "\n";
  h += "ElementType = MET_FLOAT\nElementDataFile = LOCAL\n";
  fwrite(h.data(), 1, h.size(), fp);
  s->data_offset = (long)h.size();
  return s;
}


// Zero bookkeeping of one plane for the final zero replacement: minimum positive value and the number of exact zeros in one
// branch-free pass (the scattered stack is mostly zeros at 1e8 histories: recording every position cost 5 ms per
// projection, more than the tracking kernel); positions only when the slice holds so few that finish patches them in place.
static constexpr size_t kFewZeros = 64;
static void note_zeros(MhaStack* s, int k, const float* plane) {
  const size_t n = (size_t)s->nx * s->ny, base = (size_t)k * n;
  if (s->zero_count.empty()) s->zero_count.assign((size_t)s->nslices, 0);
  float mn = s->min_positive;
  size_t nz = 0;
  for (size_t i = 0; i < n; ++i) {
    const float v = plane[i];
    nz += (v == 0.0f);
    mn = (v > 0.0f && v < mn) ? v : mn;
  }
  s->min_positive = mn;
  s->zero_count[(size_t)k] = (uint32_t)nz;
  if (nz > 0 && nz <= kFewZeros)
    for (size_t i = 0; i < n; ++i)
      if (plane[i] == 0.0f) s->zeros.push_back(base + i);
}

// Write slice k (any order, each slice once): a 4-D scan visits the projections grouped by respiratory state.
void mha_write_slice(MhaStack* s, int k, const float* plane) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (k < 0 || k >= s->nslices) throw Error(-3, "!!ERROR!! slice index outside " + s->path);
  if (s->have.empty()) s->have.assign((size_t)s->nslices, 0);
  if (s->have[k]) throw Error(-3, "!!ERROR!! slice written twice in " + s->path);
  const size_t n = (size_t)s->nx * s->ny, base = (size_t)k * n;
  note_zeros(s, k, plane);
  fseek(s->fp, s->data_offset + (long)(base * 4), SEEK_SET);
  if (fwrite(plane, sizeof(float), n, s->fp) != n) throw Error(-3, "!!ERROR!! short write to " + s->path);
  s->have[k] = 1;
  s->written++;
}

void mha_append(MhaStack* s, const float* plane) {
  if (!s->have.empty()) throw Error(-3, "!!ERROR!! " + s->path + " is written by slice index; append is not allowed");
  if (s->written >= s->nslices) throw Error(-3, "!!ERROR!! more planes appended to " + s->path + " than declared");
  const size_t n = (size_t)s->nx * s->ny;
  note_zeros(s, s->written, plane);
  if (fwrite(plane, sizeof(float), n, s->fp) != n) throw Error(-3, "!!ERROR!! short write to " + s->path);
  s->written++;
}

// replace_zeros: projections = np.where(projections == 0, projections[projections > 0].min(), projections)
// (projection.py:131-133).  Returns the replacement value (inf when the stack has no positive element).
float mha_finish(MhaStack* s, bool replace_zeros) {
  const float fill = s->min_positive;
  if (replace_zeros && std::isfinite(fill)) {
    // Slices that hold zeros are read, patched and written back whole (pread/pwrite, several threads: the work is
    // page-cache copies); a slice with only a handful of zeros gets them patched in place.  The zero positions were
    // recorded while the planes were written.
    const size_t n = (size_t)s->nx * s->ny;
    if (s->zero_count.empty()) s->zero_count.assign((size_t)s->nslices, 0);
    const std::vector<uint32_t>& count = s->zero_count;
    std::vector<size_t> first((size_t)s->nslices + 1, 0);  // positions are recorded for the few-zero slices only
    std::sort(s->zeros.begin(), s->zeros.end());           // slices written by index arrive in any order
    for (int k = 0; k < s->nslices; ++k) first[(size_t)k + 1] = first[(size_t)k] + (count[k] <= kFewZeros ? count[k] : 0);
    fflush(s->fp);
    const int fd = fileno(s->fp);
    std::vector<int> dirty;
    for (int k = 0; k < s->nslices; ++k)
      if (count[k] > 0) dirty.push_back(k);
    const int nthreads = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::thread::hardware_concurrency(), dirty.size()}));
    std::vector<int> failed((size_t)nthreads, 0);
    auto work = [&](int t) {
      std::vector<float> buf;
      for (size_t d = (size_t)t; d < dirty.size(); d += (size_t)nthreads) {
        const int k = dirty[d];
        const off_t at = (off_t)(s->data_offset + (long)((size_t)k * n * 4));
        if (count[k] <= kFewZeros) {
          for (size_t z = first[(size_t)k]; z < first[(size_t)k + 1]; ++z)
            if (pwrite(fd, &fill, 4, (off_t)(s->data_offset + (long)(s->zeros[z] * 4))) != 4) failed[(size_t)t] = 1;
          continue;
        }
        buf.resize(n);
        if (pread(fd, buf.data(), n * 4, at) != (ssize_t)(n * 4)) continue;  // slice never written: reported below
        for (float& v : buf)
          if (v == 0.0f) v = fill;
        if (pwrite(fd, buf.data(), n * 4, at) != (ssize_t)(n * 4)) failed[(size_t)t] = 1;
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    for (int f : failed)
      if (f) throw Error(-3, "!!ERROR!! short write to " + s->path);
  }
  const bool complete = s->written == s->nslices;
  const std::string path = s->path;
  fclose(s->fp);
  delete s;
  if (!complete) throw Error(-3, "!!ERROR!! " + path + " closed with fewer planes than declared");
  return fill;
}

// Minimal reader for the files written above (and by SimpleITK with the same layout): dims + float32 data.
void mha_read(const std::string& path, int dims[3], std::vector<float>& data) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) throw Error(-1, "!!ERROR!! can not open " + path);
  char line[512];
  dims[0] = dims[1] = dims[2] = 1;
  bool is_float = false, local = false;
  while (fgets(line, sizeof line, fp)) {
    if (!strncmp(line, "DimSize", 7)) {
      const char* eq = strchr(line, '=');
      if (eq) sscanf(eq + 1, "%d %d %d", &dims[0], &dims[1], &dims[2]);
    } else if (!strncmp(line, "ElementType", 11)) is_float = strstr(line, "MET_FLOAT") != nullptr && strstr(line, "MET_FLOAT_") == nullptr;
    else if (!strncmp(line, "CompressedData", 14) && strstr(line, "True")) { fclose(fp); throw Error(-2, "!!ERROR!! compressed MetaImage not supported: " + path); }
    else if (!strncmp(line, "ElementDataFile", 15)) { local = strstr(line, "LOCAL") != nullptr; break; }
  }
  if (!is_float || !local) { fclose(fp); throw Error(-2, "!!ERROR!! " + path + ": expected a MET_FLOAT MetaImage with local data"); }
  const size_t n = (size_t)dims[0] * dims[1] * dims[2];
  data.resize(n);
  const size_t got = fread(data.data(), 4, n, fp);
  fclose(fp);
  if (got != n) throw Error(-2, "!!ERROR!! " + path + ": truncated data");
}

// ------------------------------------------------------------------------------------------------
// scipy.ndimage.gaussian_filter(img, sigma=(sigma_y, sigma_x)) for a float32 image [ny][nx]: truncate = 4, mode
// 'reflect', separable correlate1d passes (axis 0 first, then axis 1), double accumulation in scipy's symmetric-
// kernel order, float32 storage between the passes (projection.py:103-105).
// ------------------------------------------------------------------------------------------------
static void gaussian_weights(double sigma, std::vector<double>& w, int& radius) {
  radius = (int)(4.0 * sigma + 0.5);
  w.resize(2 * radius + 1);
  double sum = 0.0;
  const double s2 = sigma * sigma;
  for (int i = -radius; i <= radius; ++i) { w[i + radius] = exp(-0.5 / s2 * (double)(i * i)); sum += w[i + radius]; }
  for (double& v : w) v /= sum;
}
static inline int reflect_index(int i, int n) {  // d c b a | a b c d | d c b a
  if (n == 1) return 0;
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}
static void correlate_line(const double* ext /* centre at ext[radius + k] */, int n, const std::vector<double>& w, int radius, float* out, size_t stride) {
  for (int k = 0; k < n; ++k) {
    const double* c = ext + radius + k;
    double tmp = c[0] * w[radius];
    for (int j = -radius; j < 0; ++j) tmp += (c[j] + c[-j]) * w[j + radius];
    out[(size_t)k * stride] = (float)tmp;
  }
}
void gaussian_filter_2d(float* img, int ny, int nx, double sigma_y, double sigma_x) {
  std::vector<double> w;
  int r;
  if (sigma_y > 1e-15) {
    gaussian_weights(sigma_y, w, r);
    std::vector<double> ext(ny + 2 * r);
    for (int x = 0; x < nx; ++x) {
      for (int k = -r; k < ny + r; ++k) ext[k + r] = (double)img[(size_t)reflect_index(k, ny) * nx + x];
      correlate_line(ext.data(), ny, w, r, img + x, (size_t)nx);
    }
  }
  if (sigma_x > 1e-15) {
    gaussian_weights(sigma_x, w, r);
    std::vector<double> ext(nx + 2 * r);
    for (int y = 0; y < ny; ++y) {
      float* row = img + (size_t)y * nx;
      for (int k = -r; k < nx + r; ++k) ext[k + r] = (double)row[reflect_index(k, nx)];
      correlate_line(ext.data(), nx, w, r, row, 1);
    }
  }
}

// projections_total_normalized.mha = log(gaussian_filter(air, sigma) / total) (simulation.py:258-270, projection.py:96-115),
// streamed over the planes of an existing total stack.  sigma <= 0 skips the filter.
void normalize_stack(const std::string& total_path, const std::string& air_path, double sigma_y, double sigma_x, const std::string& out_path,
                     double sx, double sy) {
  int ad[3], td[3];
  std::vector<float> air, plane;
  mha_read(air_path, ad, air);
  FILE* fp = fopen(total_path.c_str(), "rb");
  if (!fp) throw Error(-1, "!!ERROR!! can not open " + total_path);
  char line[512];
  td[0] = td[1] = td[2] = 1;
  while (fgets(line, sizeof line, fp)) {
    if (!strncmp(line, "DimSize", 7)) { const char* eq = strchr(line, '='); if (eq) sscanf(eq + 1, "%d %d %d", &td[0], &td[1], &td[2]); }
    if (!strncmp(line, "ElementDataFile", 15)) break;
  }
  if (ad[0] != td[0] || ad[1] != td[1]) { fclose(fp); throw Error(-2, "!!ERROR!! air projection and projection stack differ in size"); }
  // the air stack holds one projection (simulation.py:259-261 squeezes it)
  air.resize((size_t)ad[0] * ad[1]);
  if (sigma_y > 0.0 || sigma_x > 0.0) gaussian_filter_2d(air.data(), ad[1], ad[0], sigma_y, sigma_x);
  MhaStack* out = mha_create(out_path, td[0], td[1], td[2], sx, sy);
  const size_t n = (size_t)td[0] * td[1];
  plane.resize(n);
  std::vector<float> res(n);
  try {
    for (int k = 0; k < td[2]; ++k) {
      if (fread(plane.data(), 4, n, fp) != n) throw Error(-2, "!!ERROR!! " + total_path + ": truncated data");
      for (size_t i = 0; i < n; ++i) res[i] = logf(air[i] / plane[i]);
      mha_append(out, res.data());
    }
  } catch (...) {
    fclose(fp);
    fclose(out->fp);
    delete out;
    throw;
  }
  fclose(fp);
  mha_finish(out, false);
}

}  // namespace mcgpu
