// report.cpp -- projection ASCII writer and voxel-file writer (wire formats of the reference).
//
//  * write_projection_ascii(): the per-projection output file of the reference engine
//    (report_image, docker/mcgpu/MC-GPU_v1.3.cu:2783-2953): 20 comment lines,
//    Nz blocks of Nx lines "%.8lf %.8lf %.8lf %.8lf", a blank line after each block, 6 footer comment
//    lines.  This is what cbctmc/mc/projection.py:42 parses with np.loadtxt.  The reference prints
//    with one fprintf per pixel on rank 0 (~0.9 s per 1848x768 projection); here rows are formatted
//    by a pool of threads with an exact fixed-point formatter (falls back to snprintf when the
//    8th decimal is within rounding doubt), so the bytes are identical to "%.8lf".
//  * write_voxel_file(): the `.vox(.gz)` body written by cbctmc/mc/voxel_data.pyx:12-72 ("<mat> <dens:.6f>\n",
//    blank line after each x-row, a second blank line after each z-slice) under the header fields of
//    cbctmc/assets/templates/mcgpu_geometry.jinja2:66-73.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "host_model.hpp"

namespace mcgpu {
namespace {

inline double rad2deg(double x) { return x * 180.0 / kPi; }

// Append value formatted as "%.8lf".  Exact for the fast path: v in [0, 2^52/1e8).
inline char* put_fixed8(char* out, double v) {
  if (v >= 0.0 && v < 4.0e7) {
    const double ip = floor(v);
    const double f8 = (v - ip) * 1.0e8;  // (v-ip) exact; product within ~1.5e-8 of the true value
    const double fl = floor(f8);
    const double rem = f8 - fl;
    if (fabs(rem - 0.5) > 1.0e-6) {
      uint64_t frac = (uint64_t)fl + (rem > 0.5 ? 1u : 0u);
      uint64_t iv = (uint64_t)ip;
      if (frac >= 100000000ull) { frac -= 100000000ull; ++iv; }
      // two digits per step from a 200-byte table
      static const char kPairs[201] =
          "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
          "707172737475767778798081828384858687888990919293949596979899";
      if (iv < 10) *out++ = (char)('0' + iv);
      else {
        char tmp[24];
        int n = 0;
        while (iv >= 100) { const unsigned r = (unsigned)(iv % 100); iv /= 100; tmp[n++] = kPairs[2 * r + 1]; tmp[n++] = kPairs[2 * r]; }
        if (iv >= 10) { tmp[n++] = kPairs[2 * iv + 1]; tmp[n++] = kPairs[2 * iv]; }
        else tmp[n++] = (char)('0' + iv);
        while (n) *out++ = tmp[--n];
      }
      *out++ = '.';
      const unsigned hi = (unsigned)(frac / 10000), lo = (unsigned)(frac % 10000);
      const unsigned a = hi / 100, b = hi % 100, c = lo / 100, d = lo % 100;
      memcpy(out, kPairs + 2 * a, 2); memcpy(out + 2, kPairs + 2 * b, 2); memcpy(out + 4, kPairs + 2 * c, 2); memcpy(out + 6, kPairs + 2 * d, 2);
      return out + 8;
    }
  }
  return out + sprintf(out, "%.8lf", v);
}

}  // namespace

std::string projection_file_name(const HostModel& m, int p) {
  const SimConfig& c = m.cfg;
  float seq;
  if (c.enable_specific_angles == 0) seq = (float)rad2deg(c.initial_angle + p * c.D_angle);
  else seq = c.specific_angles[p];
  char buf[512];
  snprintf(buf, sizeof buf, "%s_%010.6fdeg", c.file_output.c_str(), seq);
  return buf;
}

namespace {
auto appendf = [](std::string& to, const char* fmt, auto... args) {
  char line[1024];
  const int k = snprintf(line, sizeof line, fmt, args...);
  to.append(line, (size_t)std::min<int>(std::max(k, 0), (int)sizeof line - 1));
};

// the 20 comment lines in front of the data (report_image, MC-GPU_v1.3.cu:2818-2858)
std::string projection_header(const HostModel& m, int p) {
  const SimConfig& c = m.cfg;
  const DetectorPose& d0 = m.detector[0];
  const int nx = d0.nx, nz = d0.nz;
  float cur, seq;
  if (c.enable_specific_angles == 0) {
    cur = (float)rad2deg(c.initial_angle + p * c.D_angle);
    seq = cur;
    if (cur >= (360 - 0.0001)) cur -= 360;
  } else {
    cur = c.specific_angles[p];
    seq = cur;
  }
  std::string header;
  const SourcePose& s = m.source[p];
  header += "# \n";
  header += "#     *****************************************************************************\n";
  header += "#     ***   MC CBCT projection engine for AMD MI355X (MC-GPU v1.3 file contract)  ***\n";
  header += "#     ***                                                                       ***\n";
  header += "#     ***   drop-in for cbctmc.mc: same inputs, same per-projection output files  ***\n";
  header += "#     *****************************************************************************\n";
  header += "# \n";
  header += "#  *** SIMULATION IN THE GPU USING HIP ***\n";
  header += "#\n";
  header += "#  Image created counting the energy arriving at each pixel: ideal energy integrating detector.\n";
  header += "#  Pixel value units: eV/cm^2 per history (energy fluence).\n";
  appendf(header, "#  CT projection %d of %d: angle from X axis = %lf (mod 360deg), %lf (no mod 360deg) \n", p + 1, c.num_projections,
                   (double)cur, (double)seq);
  appendf(header, "#  Focal spot position = (%.8f,%.8f,%.8f), cone beam direction = (%.8f,%.8f,%.8f)\n", s.pos[0], s.pos[1], s.pos[2],
                   s.dir[0], s.dir[1], s.dir[2]);
  header += c.enable_specific_angles == 0 ? "#  Specific angles enabled: NO\n" : "#  Specific angles enabled: YES\n";
  appendf(header, "#  Pixel size:  %lf x %lf = %lf cm^2\n", 1.0 / (double)d0.inv_pixel_size_X, 1.0 / (double)d0.inv_pixel_size_Z,
                   1.0 / (double)(d0.inv_pixel_size_X * d0.inv_pixel_size_Z));
  appendf(header, "#  Number of pixels in X and Z:  %d  %d\n", nx, nz);
  header += "#  (X rows given first, a blank line separates the different Z values)\n";
  header += "# \n";
  header += "#  [NON-SCATTERED] [COMPTON] [RAYLEIGH] [MULTIPLE-SCATTING]\n";
  header += "# ==========================================================\n";

  return header;
}

// the comment lines behind the data (:2906-2947)
std::string projection_footer(const HostModel& m, double energy_integral, double maximum, long max_pixel, unsigned long long total_histories,
                              double seconds) {
  const DetectorPose& d0 = m.detector[0];
  const int nx = d0.nx;
  const double SCALE = 1.0 / 100.0f;  // 1/SCALE_eV (MC-GPU_v1.3.cu:2860)
  const double NORM = SCALE * d0.inv_pixel_size_X * d0.inv_pixel_size_Z / ((double)total_histories);
  std::string footer;
  footer += "#   *** Simulation REPORT: ***\n";
  appendf(footer, "#       Fraction of energy detected (over the mean energy of the spectrum): %.3lf%%\n",
                   100.0 * SCALE * (energy_integral / (double)total_histories) / (double)m.spectrum.mean_energy);
  appendf(footer, "#       Maximum energy detected in pixel %i: (x,y)=(%i,%i) -> pixel value = %lf eV/cm^2\n", (int)max_pixel,
                   (int)(max_pixel % nx), (int)(max_pixel / nx), NORM * maximum);
  appendf(footer, "#       Simulated x rays:    %lld\n", (long long)total_histories);
  appendf(footer, "#       Simulation time [s]: %.2f\n", seconds);
  if (seconds > 0.000001) appendf(footer, "#       Speed [x-rays/sec]:  %.2f\n\n", ((double)total_histories) / seconds);
  return footer;
}

bool pwrite_all(int fd, const char* data, size_t len, off_t where) {
  size_t done = 0;
  while (done < len) {
    const ssize_t k = pwrite(fd, data + done, len - done, where + (off_t)done);
    if (k <= 0) return false;
    done += (size_t)k;
  }
  return true;
}
}  // namespace

double projection_norm(const HostModel& m, unsigned long long total_histories) {
  const DetectorPose& d0 = m.detector[0];
  const double SCALE = 1.0 / 100.0f;
  return SCALE * d0.inv_pixel_size_X * d0.inv_pixel_size_Z / ((double)total_histories);
}

// A projection file whose data lines were formatted elsewhere (on the device: ascii_device.hip): header, the text as it is,
// footer; the text goes to the file in parallel slices at their offsets.
size_t write_projection_preformatted(const HostModel& m, int p, const char* text, size_t text_bytes, double energy_integral, double maximum,
                                     long max_pixel, unsigned long long total_histories, double seconds, const std::string& file_name,
                                     int n_threads) {
  const int fd = open(file_name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " can not be opened!!");
  struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) close(fd); } } closer{fd};
  const std::string header = projection_header(m, p), footer = projection_footer(m, energy_integral, maximum, max_pixel, total_histories, seconds);
  const size_t total = header.size() + text_bytes + footer.size();
  // ONE thread per file: concurrent write()s to one file serialise on its inode lock (measured per 63 MB file: 1 thread 7 ms,
  // 2-16 threads 16-20 ms; a shared mapping filled by 16 threads 40 ms).  Parallelism comes from writing two files at a time
  // (scan.cpp: one worker per formatter slot).
  int T = n_threads > 0 ? n_threads : 1;
  std::vector<int> bad((size_t)T, 0);
  auto put = [&](int t) {
    const size_t b0 = text_bytes * (size_t)t / (size_t)T, b1 = text_bytes * ((size_t)t + 1) / (size_t)T;
    bad[(size_t)t] = !pwrite_all(fd, text + b0, b1 - b0, (off_t)(header.size() + b0));
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(put, t);
  bool ok = pwrite_all(fd, header.data(), header.size(), 0);
  put(0);
  for (auto& x : th) x.join();
  for (int t = 0; t < T; ++t) ok = ok && !bad[(size_t)t];
  ok = ok && pwrite_all(fd, footer.data(), footer.size(), (off_t)(header.size() + text_bytes));
  closer.fd = -1;
  if (close(fd) != 0 || !ok) throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " could not be written!!");
  return total;
}

size_t write_projection_ascii(const HostModel& m, int p, const uint64_t* image, unsigned long long total_histories,
                              double seconds, const std::string& file_name, int n_threads) {
  const DetectorPose& d0 = m.detector[0];
  const int nx = d0.nx, nz = d0.nz;
  const size_t npix = (size_t)nx * nz;
  // No stdio stream and no shared buffer: header, bands and footer are formatted into memory owned by this call and
  // written at their offsets with pwrite, so concurrent calls (one scan per GPU in one process) cannot interleave.
  const int fd = open(file_name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " can not be opened!!");
  struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) close(fd); } } closer{fd};
  const std::string header = projection_header(m, p);
  std::string footer;
  const double SCALE = 1.0 / 100.0f;  // 1/SCALE_eV (MC-GPU_v1.3.cu:2860)
  const double NORM = SCALE * d0.inv_pixel_size_X * d0.inv_pixel_size_Z / ((double)total_histories);

  int T = n_threads > 0 ? n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  T = std::min(T, nz);
  if (npix < 65536) T = 1;
  // text of band t: chunks[t] is a view (pointer, length) into a buffer that is allocated uninitialised (a std::string
  // would first zero-fill its 28-bytes-per-number capacity: 160 MB per projection)
  // The buffers are recycled between calls (BandPool): fresh 10 MB allocations per band and projection cost more in page
  // faults than the formatting itself.
  struct Band { std::vector<char>* mem = nullptr; size_t len = 0; const char* data() const { return mem->data(); } size_t size() const { return len; } };
  struct BandPool {
    std::mutex mu;
    std::vector<std::vector<char>*> free_list;
    std::vector<char>* take(size_t bytes) {
      std::vector<char>* v = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!free_list.empty()) { v = free_list.back(); free_list.pop_back(); }
      }
      if (!v) v = new std::vector<char>();
      if (v->size() < bytes) v->resize(bytes);  // grows (and zero-fills) only the first time
      return v;
    }
    void give(std::vector<char>* v) { std::lock_guard<std::mutex> lk(mu); free_list.push_back(v); }
    ~BandPool() { for (auto* v : free_list) delete v; }
  };
  static BandPool pool;
  std::vector<Band> chunks(T);
  std::vector<double> integral(T, 0.0), maxval(T, -100.0);
  std::vector<long> maxpix(T, 0);
  auto work = [&](int t) {
    const int z0 = (int)((long)nz * t / T), z1 = (int)((long)nz * (t + 1) / T);
    Band& out = chunks[t];
    out.mem = pool.take((size_t)(z1 - z0) * ((size_t)nx * 4 * 28 + 1) + 64);
    char* w = out.mem->data();
    double integ = 0.0, mx = -100.0;
    long mp = 0;
    for (int j = z0; j < z1; ++j) {
      size_t pix = (size_t)j * nx;
      for (int i = 0; i < nx; ++i, ++pix) {
        const double e0 = (double)image[pix], e1 = (double)image[pix + npix], e2 = (double)image[pix + 2 * npix],
                     e3 = (double)image[pix + 3 * npix];
        w = put_fixed8(w, NORM * e0); *w++ = ' ';
        w = put_fixed8(w, NORM * e1); *w++ = ' ';
        w = put_fixed8(w, NORM * e2); *w++ = ' ';
        w = put_fixed8(w, NORM * e3); *w++ = '\n';
        const double tot = e0 + e1 + e2 + e3;
        if (tot > mx) { mx = tot; mp = (long)pix; }
        integ += tot;
      }
      *w++ = '\n';
    }
    out.len = (size_t)(w - out.mem->data());
    integral[t] = integ;
    maxval[t] = mx;
    maxpix[t] = mp;
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  double energy_integral = 0.0, maximum = -100.0;
  long max_pixel = 0;
  off_t data_end = 0;
  {
    // the 63 MB of text go to the file in parallel too: every band is written at its own offset (pwrite); a serial fwrite of
    // the bands was 15 of the 20 ms a projection cost
    std::vector<off_t> at((size_t)T + 1, (off_t)header.size());
    for (int t = 0; t < T; ++t) at[(size_t)t + 1] = at[(size_t)t] + (off_t)chunks[t].size();
    std::vector<int> bad((size_t)T + 1, 0);
    auto put_at = [&](const char* data, size_t len, off_t where) {
      size_t done = 0;
      while (done < len) {
        const ssize_t k = pwrite(fd, data + done, len - done, where + (off_t)done);
        if (k <= 0) return false;
        done += (size_t)k;
      }
      return true;
    };
    auto put = [&](int t) { bad[(size_t)t] = !put_at(chunks[t].data(), chunks[t].size(), at[(size_t)t]); };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(put, t);
    bad[(size_t)T] = !put_at(header.data(), header.size(), 0);
    put(0);
    for (auto& x : th) x.join();
    for (int t = 0; t <= T; ++t)
      if (bad[(size_t)t]) {
        for (int u = 0; u < T; ++u) pool.give(chunks[u].mem);
        throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " could not be written!!");
      }
    data_end = at[(size_t)T];
  }
  size_t bytes = header.size();
  for (int t = 0; t < T; ++t) {
    pool.give(chunks[t].mem);
    bytes += chunks[t].size();
    energy_integral += integral[t];  // NB: summed per band; the footer's %.3lf is insensitive to the order
    if (maxval[t] > maximum) { maximum = maxval[t]; max_pixel = maxpix[t]; }
  }
  footer = projection_footer(m, energy_integral, maximum, max_pixel, total_histories, seconds);
  {
    size_t done = 0;
    while (done < footer.size()) {
      const ssize_t k = pwrite(fd, footer.data() + done, footer.size() - done, data_end + (off_t)done);
      if (k <= 0) throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " could not be written!!");
      done += (size_t)k;
    }
    bytes += footer.size();
  }
  closer.fd = -1;
  if (close(fd) != 0) throw Error(-3, "!!fopen ERROR report_image!! File " + file_name + " could not be written!!");
  return bytes;
}

// ---------------------------------------------------------------------------------------------
size_t write_voxel_file(const std::string& path, const int n[3], const float spacing_cm[3], const uint8_t* material,
                        const float* density, bool gzip) {
  const size_t nx = n[0], ny = n[1], nz = n[2];
  std::string head;
  char buf[512];
  head += "# MC-GPU voxelized geometry (penEasy 2008 voxel format); x runs first, then y, then z\n";
  head += "[SECTION VOXELS HEADER v.2008-04-13]\n";
  snprintf(buf, sizeof buf, "%d %d %d  # SIZE IN X, Y, Z\n", n[0], n[1], n[2]);
  head += buf;
  snprintf(buf, sizeof buf, "%.9g %.9g %.9g  # VOXEL SPACING IN X, Y, Z\n", spacing_cm[0], spacing_cm[1], spacing_cm[2]);
  head += buf;
  head += "1  # COLUMN NUMBER WHERE MATERIAL ID IS LOCATED\n2  # COLUMN NUMBER WHERE MASS DENSITY IS LOCATED\n"
          "1  # BLANK LINES AT END OF X,Y-CYCLES (1=YES, 0=NO)\n[END OF VXH SECTION]\n#\n# >>>> DATA BEGINS >>>>\n";
  gzFile gz = nullptr;
  FILE* fp = nullptr;
  if (gzip) {
    gz = gzopen(path.c_str(), "wb1");
    if (!gz) throw Error(-3, "!!ERROR!! Voxel file " + path + " can not be opened for writing!!");
    gzbuffer(gz, 1 << 20);
  } else {
    fp = fopen(path.c_str(), "wb");
    if (!fp) throw Error(-3, "!!ERROR!! Voxel file " + path + " can not be opened for writing!!");
  }
  size_t bytes = 0;
  auto emit = [&](const char* p, size_t len) {
    if (gz) gzwrite(gz, p, (unsigned)len); else fwrite(p, 1, len, fp);
    bytes += len;
  };
  emit(head.data(), head.size());
  // format one z-slice at a time, rows in parallel
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::string> rows(T);
  for (size_t k = 0; k < nz; ++k) {
    auto work = [&](int t) {
      const size_t j0 = ny * t / T, j1 = ny * (t + 1) / T;
      std::string& out = rows[t];
      out.resize((j1 - j0) * (nx * 24 + 1) + 8);
      char* w = &out[0];
      for (size_t j = j0; j < j1; ++j) {
        const size_t base = (k * ny + j) * nx;
        for (size_t i = 0; i < nx; ++i) {
          unsigned mv = material[base + i];
          if (mv >= 100) { *w++ = (char)('0' + mv / 100); mv %= 100; *w++ = (char)('0' + mv / 10); *w++ = (char)('0' + mv % 10); }
          else if (mv >= 10) { *w++ = (char)('0' + mv / 10); *w++ = (char)('0' + mv % 10); }
          else *w++ = (char)('0' + mv);
          *w++ = ' ';
          // "%.6f" of the float density (python f'{x:.6f}' formats the float promoted to double)
          const double dv = (double)density[base + i];
          const double ip = floor(dv);
          const double f6 = (dv - ip) * 1.0e6;
          const double fl = floor(f6), rem = f6 - fl;
          if (dv >= 0.0 && dv < 1.0e9 && fabs(rem - 0.5) > 1.0e-6) {
            uint64_t frac = (uint64_t)fl + (rem > 0.5 ? 1u : 0u), iv = (uint64_t)ip;
            if (frac >= 1000000ull) { frac -= 1000000ull; ++iv; }
            char tmp[24];
            int nn = 0;
            do { tmp[nn++] = (char)('0' + iv % 10); iv /= 10; } while (iv);
            while (nn) *w++ = tmp[--nn];
            *w++ = '.';
            for (int q = 5; q >= 0; --q) { w[q] = (char)('0' + frac % 10); frac /= 10; }
            w += 6;
          } else {
            w += sprintf(w, "%.6f", dv);
          }
          *w++ = '\n';
        }
        *w++ = '\n';
      }
      out.resize((size_t)(w - &out[0]));
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; ++t) emit(rows[t].data(), rows[t].size());
    emit("\n", 1);
  }
  if (gz) gzclose(gz); else fclose(fp);
  return bytes;
}


// ---------------------------------------------------------------------------------------------
// Binary sidecar (format: magic "MCGVOX1\n", u32 version, u32 kind {0 raw, 1 palette+u8, 2 palette+u16}, i32 n[3],
// f32 voxel_size[3], u32 palette_count, palette {u32 material, f32 density}..., then the voxel data, x fastest)
// ---------------------------------------------------------------------------------------------
std::string voxel_sidecar_path(const std::string& voxel_file) {
  std::string s = voxel_file;
  auto strip = [&](const char* ext) {
    const size_t n = strlen(ext);
    if (s.size() > n && s.compare(s.size() - n, n, ext) == 0) s.resize(s.size() - n);
  };
  strip(".gz");
  strip(".vox");
  return s + ".voxbin";
}

size_t write_voxel_binary(const std::string& path, const int n[3], const float spacing_cm[3], const uint8_t* material,
                          const float* density) {
  const size_t nvox = (size_t)n[0] * n[1] * n[2];
  // densities as the text file carries them: "%.6f", read back with %f (voxel_data.pyx:25, MC-GPU_v1.3.cu:2117)
  std::unordered_map<uint32_t, float> quantised;
  auto q6 = [&](float d) {
    uint32_t b;
    memcpy(&b, &d, 4);
    auto it = quantised.find(b);
    if (it != quantised.end()) return it->second;
    char t[64];
    snprintf(t, sizeof t, "%.6f", (double)d);
    const float v = strtof(t, nullptr);
    quantised.emplace(b, v);
    return v;
  };
  std::unordered_map<uint64_t, uint32_t> index_of;
  std::vector<uint32_t> pal_mat;
  std::vector<float> pal_dens;
  std::vector<uint16_t> idx(nvox);
  std::vector<float> dq(nvox);
  bool overflow = false;
  uint64_t last_key = ~0ull;
  uint32_t last_idx = 0;
  for (size_t i = 0; i < nvox; ++i) {
    const float d = q6(density[i]);
    dq[i] = d;
    if (overflow) continue;
    uint32_t db;
    memcpy(&db, &d, 4);
    const uint64_t key = ((uint64_t)material[i] << 32) | db;
    if (key != last_key) {
      auto it = index_of.find(key);
      if (it == index_of.end()) {
        if (index_of.size() >= 65536) { overflow = true; continue; }
        last_idx = (uint32_t)index_of.size();
        index_of.emplace(key, last_idx);
        pal_mat.push_back(material[i]);
        pal_dens.push_back(d);
      } else last_idx = it->second;
      last_key = key;
    }
    idx[i] = (uint16_t)last_idx;
  }
  const uint32_t kind = overflow ? 0u : (index_of.size() <= 256 ? 1u : 2u);
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) throw Error(-3, "!!ERROR!! can not open " + path + " for writing");
  size_t bytes = 0;
  auto put = [&](const void* p, size_t nb) { bytes += fwrite(p, 1, nb, fp); };
  const uint32_t version = 1, npal = overflow ? 0u : (uint32_t)index_of.size();
  put("MCGVOX1\n", 8);
  put(&version, 4); put(&kind, 4); put(n, 12); put(spacing_cm, 12); put(&npal, 4);
  for (uint32_t k = 0; k < npal; ++k) { put(&pal_mat[k], 4); put(&pal_dens[k], 4); }
  if (kind == 0) { put(material, nvox); put(dq.data(), nvox * 4); }
  else if (kind == 1) {
    std::vector<uint8_t> i8(nvox);
    for (size_t i = 0; i < nvox; ++i) i8[i] = (uint8_t)idx[i];
    put(i8.data(), nvox);
  } else put(idx.data(), nvox * 2);
  if (fclose(fp) != 0) throw Error(-3, "!!ERROR!! " + path + " could not be written");
  return bytes;
}

}  // namespace mcgpu
