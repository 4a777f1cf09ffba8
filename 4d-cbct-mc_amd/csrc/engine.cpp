// engine.cpp -- device upload, launch and the C ABI (include/mcgpu_amd.h) of the MC CBCT engine.
//
// Replaces init_CUDA_device (docker/mcgpu/MC-GPU_v1.3.cu:2454-2724) and the per-projection driver of
// main() (:667-1056).  Compiled with hipcc; every HIP call lives here or in the kernel TUs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>

#include "../../include/mcgpu_amd.h"
#include "device_model.hpp"
#include "ascii_device.hpp"
#include "geometry_device.hpp"

namespace mcgpu {

hipError_t launch_track_compat(const TrackArgs& args, int blocks, hipStream_t stream);
hipError_t launch_track_fast(const TrackArgs& args, int blocks, hipStream_t stream);
int occupancy_track_fast(const TrackArgs& args);
#if defined(MC_WITH_STATS) && MC_WITH_STATS
hipError_t launch_track_stats(const TrackArgs& args, int blocks, hipStream_t stream);  // diagnostic library only (track_stats.o)
#endif
hipError_t microbench_valu_issue(int num_cus, double out3[3], hipStream_t stream);
hipError_t microbench_atomic_rate(double* out, hipStream_t stream);
hipError_t launch_kat_rng(int mode, int seed, int batch, int hpt, int n, float* out_dev, hipStream_t stream);
hipError_t launch_kat_streams_fast(int generator, unsigned int seed, unsigned int stream_key, unsigned long long first_id,
                                   const unsigned long long* ids_dev, int n_ids, int n_draws, unsigned int* out_dev, hipStream_t stream);
hipError_t launch_kat_math(int n, const double* x, double* l, double* e, double* s, double* c, hipStream_t stream);
hipError_t launch_kat_expf(int n, const float* x, float* e, hipStream_t stream);
hipError_t launch_kat_f32(int op, int n, const float* a, const float* b, float* out, hipStream_t stream);
hipError_t launch_warp(int nx, int ny, int nz, const unsigned char* mat, const float* dens, const float* dvf, unsigned char default_mat,
                       float default_dens, unsigned char* out_mat, float* out_dens, hipStream_t stream);
hipError_t launch_finalize(unsigned long long* image, int nx, int nz, int crop_nx, double norm, float* planes, int clear, hipStream_t stream);

namespace {

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
  g_last_error = msg.find("ERROR") == std::string::npos ? "!!ERROR!! " + msg : msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) throw Error(-1, std::string("!!HIP ERROR!! ") + #expr + ": " + hipGetErrorString(_e)); \
  } while (0)

struct DeviceModel {
  int device_id = -1;
  void* vol = nullptr;
  size_t vol_bytes = 0;
  int vol_kind = kVolU8, palette_size = 0;
  float* palette = nullptr;
  unsigned char* bricks = nullptr;
  // on-device formatter of the ASCII projection files (mcgpu_format_projection): a few slots, so that the host writes
  // the text of earlier projections while the next is formatted
  struct AsciiSlot {
    char* text_dev = nullptr;
    char* text_host = nullptr;            // pinned
    unsigned long long* rows_dev = nullptr;   // row_len[nz] row_off[nz+1] row_arg[nz] | row_sum[nz] row_max[nz] | flags
    unsigned long long* rows_host = nullptr;  // pinned copy of the same block
    hipStream_t copy_stream = nullptr;        // the slot's download (copy engine)
    hipEvent_t ready = nullptr;               // recorded behind the formatter and the download of the row block
  } ascii[MCGPU_ASCII_SLOTS];
  unsigned long long ascii_capacity = 0;
  // on-device geometry changes (mcgpu_warp_geometry): the base geometry's palette index volume, scratch, the palette on the
  // host and the code assignment of the base geometry
  unsigned char* vol_base = nullptr;
  unsigned short *sub_first = nullptr, *brick_first = nullptr;
  unsigned char* code_of_dev = nullptr;
  unsigned int* rebuild_out = nullptr;
  float* dvf = nullptr;
  std::vector<float> palette_host;  // {density, bits(compact material)} pairs
  unsigned char code_of[256];
  int background = 0;
  unsigned char* sub = nullptr;   // second-level codes: 4 bits per sub-brick of 4^3 voxels, dense over the volume (u8 volumes)
  int sub_n[3] = {1, 1, 1}, sub_mixed = 0;
  int brick_shift = 0, brick_n[3] = {1, 1, 1}, brick_count = 0, brick_bytes = 0, bricks_mixed = 0;
  int brick_palette[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int has_exterior = 0, bricks_exterior = 0;
  float objbox_lo[3] = {0, 0, 0}, objbox_hi[3] = {0, 0, 0};
  int num_spectrum_bins = 0;
  int shell_first[kMaxMaterials] = {0};
  LdsLayout lds;
  TrackCold* cold = nullptr;      // device copy of the rarely used table pointers
  TrackCold cold_host;            // its host image (re-uploaded when a tuning knob changes)
  unsigned long long* dose_voxels = nullptr;     // ulonglong2 per ROI voxel (null: tally off)
  unsigned long long* dose_materials = nullptr;  // ulonglong2 x 25 (null: tally off)
  size_t dose_roi_voxels = 0;
  int dose_flags = 0;
  SourcePose* src_all = nullptr;  // [num_projections]
  DetectorPose* det_all = nullptr;
  int resident_fast = 0;  // workgroups per CU (occupancy query), 0 = not asked yet
  unsigned long long* stats = nullptr;  // kNumStats scheduler counters of the diagnostic build
  unsigned long long* work_counter = nullptr;  // history-id dispenser of the FAST kernel
  unsigned long long* scratch_image = nullptr;  // device tally of mcgpu_run_projection (allocated on first use)
  float *woodcock = nullptr, *mfp = nullptr, *mfp_tot = nullptr;
  unsigned short* sig_mid = nullptr;  // cross-section brackets (FAST flight step), see upload_model
  float* sig_w = nullptr;
  int sig_shift = -1, sig_coarse = 0;
  int sched[5] = {32, 8, 36, 12, 40};  // FAST batching thresholds {compton, rayleigh, new, flyable_low, swap_batch} (mcgpu_set_fast_schedule)
  // Tuning knobs of the environment.  Read when the device model is built and again only by
  // mcgpu_reload_env_knobs: the launch path itself never looks at the environment and never synchronises.
  struct Knobs {
    int exterior_mode = 3;                           // MCGPU_EXTERIOR_MODE: bit 0 hop during flight, bit 1 hop at the source
    bool compat_stats = false;                       // MCGPU_COMPAT_STATS: hand the diagnostic COMPAT build its counter buffer
    int compat_thresh[4] = {-1, -1, -1, -1};         // MCGPU_COMPAT_THRESH_{COMPTON,RAYLEIGH,NEW,TAKE}; -1: chosen from the materials (make_args)
    int blocks_per_cu = 0;                           // MCGPU_BLOCKS_PER_CU (0: ask the occupancy API)
    int grid_spare_percent = 0;                      // MCGPU_GRID_SPARE_PERCENT
    int sched_override[5] = {-1, -1, -1, -1, -1};    // MCGPU_THRESH_{COMPTON,RAYLEIGH,NEW}, MCGPU_FLYABLE_LOW, MCGPU_SWAP_BATCH (-1: sched[])
    int slot_trade = 3, hold_q = 6;                  // MCGPU_SLOT_TRADE, MCGPU_HOLD_Q
    bool no_exterior = false;                        // MCGPU_NO_EXTERIOR (also read by the geometry builders)
  } knobs;
  std::vector<float> sig_tot_host;    // copy of mfp_tot for the bracket builder
  float *xco = nullptr, *pco = nullptr, *aco = nullptr, *bco = nullptr;
  unsigned char *itl = nullptr, *itu = nullptr;
  float *fco = nullptr, *uico = nullptr, *fj0 = nullptr;
  float* s0_bounds = nullptr;  // COMPAT: TrackCold::s0_bounds
  float s0_emin = 0.f, s0_inv_w = 0.f;
  int* noscco = nullptr;
  float* shell_cut = nullptr;      // FAST: alias table of the Compton shell weights
  unsigned char* shell_alias = nullptr;
  float *espc = nullptr, *cutoff = nullptr;
  short* alias = nullptr;
  int nmat = 0;
  int compact_of[kMaxMaterials];
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool timed = false;
  int num_cus = 256;
  std::vector<void*> allocations;

  template <typename T>
  T* put(const std::vector<T>& host) {
    void* d = nullptr;
    const size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(&d, bytes));
    allocations.push_back(d);
    if (!host.empty()) HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return (T*)d;
  }
  void release() {
    for (void* p : allocations) (void)hipFree(p);
    allocations.clear();
    for (AsciiSlot& a : ascii) {
      if (a.text_host) (void)hipHostFree(a.text_host);
      if (a.rows_host) (void)hipHostFree(a.rows_host);
      if (a.copy_stream) (void)hipStreamDestroy(a.copy_stream);
      if (a.ready) (void)hipEventDestroy(a.ready);
      a = AsciiSlot();
    }
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (ev_stop) (void)hipEventDestroy(ev_stop);
    ev_start = ev_stop = nullptr;
  }
};

}  // namespace
}  // namespace mcgpu

struct mcgpu_ctx {
  mcgpu::HostModel host;
  mcgpu::DeviceModel dev;
  bool has_device = false;
  bool host_voxels_stale = false;  // the device holds a geometry warped there (mcgpu_warp_geometry): H.voxels is downloaded on demand
  std::map<std::string, std::vector<unsigned char>> table_cache;
};

namespace mcgpu {
namespace {

void read_env_knobs(DeviceModel& D) {
  auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
  DeviceModel::Knobs k;
  k.exterior_mode = env_int("MCGPU_EXTERIOR_MODE", 3);
  k.compat_thresh[0] = env_int("MCGPU_COMPAT_THRESH_COMPTON", -1);
  k.compat_thresh[1] = env_int("MCGPU_COMPAT_THRESH_RAYLEIGH", -1);
  k.compat_thresh[2] = env_int("MCGPU_COMPAT_THRESH_NEW", -1);
  k.compat_thresh[3] = env_int("MCGPU_COMPAT_THRESH_TAKE", -1);
  k.compat_stats = env_int("MCGPU_COMPAT_STATS", 0) != 0;
  k.blocks_per_cu = std::max(0, env_int("MCGPU_BLOCKS_PER_CU", 0));
  k.grid_spare_percent = std::max(0, env_int("MCGPU_GRID_SPARE_PERCENT", 0));
  static const char* const kSched[5] = {"MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW", "MCGPU_FLYABLE_LOW", "MCGPU_SWAP_BATCH"};
  for (int i = 0; i < 5; ++i) k.sched_override[i] = env_int(kSched[i], -1);
  k.slot_trade = env_int("MCGPU_SLOT_TRADE", 3);
  k.hold_q = env_int("MCGPU_HOLD_Q", 6) & 15;
  k.no_exterior = getenv("MCGPU_NO_EXTERIOR") != nullptr;
  D.knobs = k;
}

// The FAST scheduler's parameters live in TrackCold (device memory read through the scalar cache): effective value =
// environment override, else the schedule set by mcgpu_set_fast_schedule.  Uploads only when something changed, after the
// device has drained (callers are set-up paths, never a launch).
void apply_schedule(DeviceModel& D) {
  if (!D.cold) return;
  TrackCold& ch = D.cold_host;
  int want[5];
  for (int i = 0; i < 5; ++i) want[i] = D.knobs.sched_override[i] >= 0 ? D.knobs.sched_override[i] : D.sched[i];
  want[3] = std::max(1, want[3]);
  want[4] = std::max(1, want[4]);
  // bit 0: slots traded before flight, bit 1: before the Compton and tally/source services; bits 8-11: hold_q (sixteenths
  // of the flying lanes that end a flight segment at the latest)
  const int trade = D.knobs.slot_trade | (D.knobs.hold_q << 8);
  if (ch.trade_slots == trade && ch.thresh_compton == want[0] && ch.thresh_rayleigh == want[1] && ch.thresh_new == want[2] && ch.flyable_low == want[3] &&
      ch.swap_batch == want[4])
    return;
  ch.thresh_compton = want[0]; ch.thresh_rayleigh = want[1]; ch.thresh_new = want[2]; ch.flyable_low = want[3]; ch.swap_batch = want[4];
  ch.trade_slots = trade;
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(D.cold, &ch, sizeof ch, hipMemcpyHostToDevice));
}

// COMPAT kernel: bounds of S0(E) = sum_i f_i n_i(E, theta = pi) (K.cu:1300-1314), the normalisation of the Compton angle test
// "xi S0 > S(theta) T(tau)" (K.cu:1367-1372).  Computing it costs a second pass over the 29-40 electron shells of a tissue per
// event -- 46 % of the kernel's Compton work -- and the test rarely needs its exact value.  S0 does not decrease with E: a
// shell enters at E > U_i with a positive term, and each term grows with E (p_z(theta = pi) grows with E (E - U_i), the
// profile integral n_i with p_z).  So S0 at the lower / upper edge of an energy bin bounds it inside the bin.  The edges are
// evaluated here in double, one bin of slack on either side absorbs the float rounding of the kernel's bin index, and a
// relative margin of 1e-4 covers the float arithmetic of the reference's own S0 (terms accurate to ~2e-6 of f_i, forty
// additions of 6e-8 each).  A test that both bounds decide alike is decided; for the rest (well below 1 % with 1024 bins) the kernel computes S0.
// Layout: float2 {lo, hi} at [row * kS0Bins + bin]; row = compact material index with `compact_of`, else material number - 1.
static std::vector<float> build_s0_bounds(const HostModel& H, const int* compact_of, int rows, float* emin_out, float* inv_w_out) {
  const double emin = H.mat.e0, emax = H.mat.e0 + (double)(H.mat.num_values - 1) / H.mat.ide, w = (emax - emin) / kS0Bins;
  const double mc2 = (double)510998.918f, c1 = (double)0.707106781186545f, c2 = (double)1.4142135623731f;
  std::vector<float> bounds((size_t)2 * kS0Bins * std::max(rows, 1), 0.f);
  for (int m = 0; m < kMaxMaterials; ++m) {
    const int mc = compact_of ? compact_of[m] : (m < rows ? m : -1);
    if (mc < 0) continue;
    const int n = std::min(H.mat.noscco[m], kMaxShells);
    double fsum = 0.0;
    for (int i = 0; i < n; ++i) fsum += (double)H.mat.fco[m + i * kMaxMaterials];
    auto s0_at = [&](double E) {
      double acc = 0.0;
      for (int i = 0; i < n; ++i) {
        const double U = H.mat.uico[m + i * kMaxMaterials], J = H.mat.fj0[m + i * kMaxMaterials], f = H.mat.fco[m + i * kMaxMaterials];
        if (!(U < E)) continue;
        const double aux = E * (E - U) * 2.0;
        const double pz = J * (aux - U * mc2) / (std::sqrt(aux + aux + U * U) * mc2);
        const double a = pz > 0.0 ? c1 + pz * c2 : c1 - pz * c2;
        const double t = 0.5 * std::exp(0.5 - a * a);
        acc += f * (pz > 0.0 ? 1.0 - t : t);
      }
      return acc;
    };
    // the monotonicity argument needs shells with f >= 0, J > 0, U >= 0 (every PENELOPE table has them); a file that breaks it
    // gets bounds that decide nothing: the kernel then computes S0 for every test, like the reference
    bool regular = true;
    for (int i = 0; i < n; ++i)
      regular = regular && H.mat.fco[m + i * kMaxMaterials] >= 0.f && H.mat.fj0[m + i * kMaxMaterials] > 0.f && H.mat.uico[m + i * kMaxMaterials] >= 0.f;
    if (!regular) {
      for (int k = 0; k < kS0Bins; ++k) {
        bounds[2 * ((size_t)mc * kS0Bins + k)] = 0.f;
        bounds[2 * ((size_t)mc * kS0Bins + k) + 1] = 3.0e38f;
      }
      continue;
    }
    std::vector<double> edge(kS0Bins + 1);
    for (int k = 0; k <= kS0Bins; ++k) edge[k] = s0_at(emin + k * w);
    for (int k = 0; k < kS0Bins; ++k) {
      const double lo = k >= 1 ? edge[k - 1] * (1.0 - 1e-4) : 0.0;
      const double hi = (k + 2 <= kS0Bins ? edge[k + 2] : fsum) * (1.0 + 1e-4);
      bounds[2 * ((size_t)mc * kS0Bins + k)] = std::nextafterf((float)lo, -1.0f);
      bounds[2 * ((size_t)mc * kS0Bins + k) + 1] = std::nextafterf((float)hi, 3.0e38f);
    }
  }
  if (emin_out) *emin_out = (float)emin;
  if (inv_w_out) *inv_w_out = (float)(1.0 / w);
  return bounds;
}

// Build the palette-compressed volume and the compact-material tables and upload everything.
void upload_model(mcgpu_ctx& C, int device_id) {
  const HostModel& H = C.host;
  DeviceModel& D = C.dev;
  HIP_TRY(hipSetDevice(device_id));
  D.device_id = device_id;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  D.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  read_env_knobs(D);

  // compact material numbering
  D.nmat = 0;
  for (int m = 0; m < kMaxMaterials; ++m) D.compact_of[m] = H.mat.used[m] ? D.nmat++ : -1;
  const int nmat = D.nmat, nv = H.mat.num_values;

  // ---- volume -> palette indices
  const size_t nvox = H.voxels.count();
  std::unordered_map<uint64_t, int> index_of;
  std::vector<float> palette;  // {density, bits(mat_c)}
  std::vector<uint16_t> idx16(nvox);
  bool overflow = false;
  {
    uint64_t last_key = ~0ull;
    int last_idx = -1;
    for (size_t i = 0; i < nvox; ++i) {
      uint32_t db;
      memcpy(&db, &H.voxels.density[i], 4);
      const uint64_t key = ((uint64_t)H.voxels.material[i] << 32) | db;
      if (key != last_key) {
        auto it = index_of.find(key);
        if (it == index_of.end()) {
          if (index_of.size() >= 65536) { overflow = true; break; }
          const int mc = D.compact_of[H.voxels.material[i] - 1];
          if (mc < 0) throw Error(-2, "!!ERROR!! A voxel uses material " + std::to_string((int)H.voxels.material[i]) + " but no data file was given for it.");
          float mcf;
          memcpy(&mcf, &mc, 4);
          last_idx = (int)index_of.size();
          index_of.emplace(key, last_idx);
          palette.push_back(H.voxels.density[i]);
          palette.push_back(mcf);
        } else {
          last_idx = it->second;
        }
        last_key = key;
      }
      idx16[i] = (uint16_t)last_idx;
    }
  }
  if (overflow) {
    D.vol_kind = kVolRaw;
    std::vector<float> raw(2 * nvox);
    for (size_t i = 0; i < nvox; ++i) {
      const int mc = D.compact_of[H.voxels.material[i] - 1];
      if (mc < 0) throw Error(-2, "!!ERROR!! A voxel uses a material without data file.");
      raw[2 * i] = H.voxels.density[i];
      memcpy(&raw[2 * i + 1], &mc, 4);
    }
    D.vol = D.put(raw);
    D.vol_bytes = raw.size() * 4;
    D.palette_size = 0;
    D.palette = D.put(std::vector<float>(2, 0.f));
  } else if (index_of.size() <= 256) {
    D.vol_kind = kVolU8;
    {
      // the reference's default for voxels warped in from outside the volume (air: material 1 at 0.0013 g/cm^3,
      // cbctmc/mc/geometry.py:403-418) gets a palette entry even when no voxel holds it yet, so that a geometry can be
      // warped on the device without touching the palette (mcgpu_warp_geometry)
      const float air = 0.0013f;
      uint32_t db;
      memcpy(&db, &air, 4);
      const uint64_t key = ((uint64_t)1 << 32) | db;
      if (!index_of.count(key) && index_of.size() < 256 && D.compact_of[0] >= 0) {
        const int mc = D.compact_of[0];
        float mcf;
        memcpy(&mcf, &mc, 4);
        index_of.emplace(key, (int)index_of.size());
        palette.push_back(air);
        palette.push_back(mcf);
      }
    }
    D.palette_host = palette;
    std::vector<uint8_t> idx8(nvox);
    for (size_t i = 0; i < nvox; ++i) idx8[i] = (uint8_t)idx16[i];
    D.palette_size = (int)index_of.size();
    D.palette = D.put(palette);
    // brick grid: smallest power-of-two brick (>= 4 voxels) that keeps the grid within the LDS budget
    const int nx = H.voxels.n[0], ny = H.voxels.n[1], nz = H.voxels.n[2];
    {
      // device layout: tiles of 4x4x4 voxels = one 64-byte sector = one sub-brick of the second level (device_model.hpp:
      // tiled_voxel); the padding voxels of edge tiles repeat the tile's first voxel and are never addressed
      const unsigned int snx = (unsigned int)((nx + 3) >> 2), sny = (unsigned int)((ny + 3) >> 2), snz = (unsigned int)((nz + 3) >> 2);
      const size_t tiles = (size_t)snx * sny * snz;
      if (tiles * 64 >= (1ULL << 31)) throw Error(-2, "!!ERROR!! voxel grid too large for the 32-bit voxel index of the kernel");
      std::vector<uint8_t> tiled(tiles * 64);
      for (size_t t = 0; t < tiles; ++t) {
        const int x0 = (int)(t % snx) << 2, y0 = (int)((t / snx) % sny) << 2, z0 = (int)(t / ((size_t)snx * sny)) << 2;
        const uint8_t pad = idx8[((size_t)z0 * ny + y0) * nx + x0];
        for (int dz = 0; dz < 4; ++dz)
          for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) {
              const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
              tiled[t * 64 + (size_t)(dz * 16 + dy * 4 + dx)] = (x < nx && y < ny && z < nz) ? idx8[((size_t)z * ny + y) * nx + x] : pad;
            }
      }
      D.vol = D.put(tiled);
      D.vol_bytes = tiled.size();
    }
    int k = 2;
    auto nb = [&](int n, int sh) { return (n + (1 << sh) - 1) >> sh; };
    const char* mb = getenv("MCGPU_MAX_BRICKS");  // tuning knob: a coarser grid frees LDS
    long max_bricks = mb ? std::min<long>(std::max<long>(atol(mb), 1), kMaxBricks) : kMaxBricks;
    {
      // The FAST kernel wants two 1024-thread workgroups per CU, i.e. an LDS image of at most 80 KB.  Everything but the
      // brick grid is fixed by the materials in use (22 tissue materials: 458 Compton shells = 7.3 KB against 1.4 KB for
      // the Catphan set), so the grid gets what is left after the tables, the history slots and a coarse bracket table.
      int shells = 0;
      for (int m = 0; m < kMaxMaterials; ++m)
        if (D.compact_of[m] >= 0) shells += std::min(H.mat.noscco[m], kMaxShells);
      const int ns = std::min(H.spectrum.num_bins + 1, kMaxSpectrumBins) + 1;
      const int nc = (nv + (1 << 9) - 1) >> 9;  // brackets no coarser than 2^9 table bins
      const long fixed = std::max(shells, 1) * 16 + std::max(nmat, 1) * 8 + ns * 10 + (16 + (long)index_of.size()) * 8 + 2 * kMaxMaterials * 8 +
                         (long)kSlotWords * kPoolParked * kPoolBlockThreads * 4 + nc * nmat * 2 + nc * 4 + 12 * 16;
      const long left = 160 * 1024 / 2 - fixed;
      if (left > 0) max_bricks = std::min(max_bricks, std::max(2 * left, 512L));
    }
    while ((long)nb(nx, k) * nb(ny, k) * nb(nz, k) > max_bricks) ++k;
    D.brick_shift = k;
    D.brick_n[0] = nb(nx, k); D.brick_n[1] = nb(ny, k); D.brick_n[2] = nb(nz, k);
    D.brick_count = D.brick_n[0] * D.brick_n[1] * D.brick_n[2];
    std::vector<int> first(D.brick_count, -1);
    std::vector<unsigned char> mixed(D.brick_count, 0);
    // second level: sub-bricks of 4^3 voxels, dense over the volume (first2: palette entry, 0x100 = mixed)
    D.sub_n[0] = (nx + 3) >> 2; D.sub_n[1] = (ny + 3) >> 2; D.sub_n[2] = (nz + 3) >> 2;
    const size_t nsub = (size_t)D.sub_n[0] * D.sub_n[1] * D.sub_n[2];
    std::vector<short> first2(nsub, -1);
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y) {
        const size_t row = ((size_t)z * ny + y) * nx;
        const size_t brow = ((size_t)(z >> k) * D.brick_n[1] + (y >> k)) * D.brick_n[0];
        const size_t srow = ((size_t)(z >> 2) * D.sub_n[1] + (y >> 2)) * D.sub_n[0];
        for (int x = 0; x < nx; ++x) {
          const int b = (int)(brow + (x >> k)), v = idx8[row + x];
          if (first[b] < 0) first[b] = v;
          else if (first[b] != v) mixed[b] = 1;
          short& f2 = first2[srow + (x >> 2)];
          if (f2 < 0) f2 = (short)v;
          else if (f2 != v) f2 = 0x100;
        }
      }
    // 4-bit codes: the 14 most frequent palette entries among homogeneous bricks get codes 0..13, every other
    // brick (mixed, or a rarer homogeneous one) is 0xF = "read the voxel"; code 14 = EXTERIOR (below)
    std::vector<long> homogeneous(256, 0);
    for (int b = 0; b < D.brick_count; ++b)
      if (!mixed[b] && first[b] >= 0) ++homogeneous[first[b]];
    std::vector<int> order(256);
    for (int i = 0; i < 256; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return homogeneous[a] > homogeneous[b]; });
    int code_of[256];
    for (int i = 0; i < 256; ++i) code_of[i] = 0xF;
    for (int c = 0; c < 14; ++c) {
      D.brick_palette[c] = 0;
      if (homogeneous[order[c]] > 0) { code_of[order[c]] = c; D.brick_palette[c] = order[c]; }
    }
    D.brick_palette[14] = D.brick_palette[15] = 0;
    for (int i = 0; i < 256; ++i) D.code_of[i] = (unsigned char)code_of[i];
    D.background = order[0];
    // Exterior: the object box is the bounding box (in bricks) of every brick that is not homogeneous background
    // (background = the most frequent homogeneous entry).  Bricks outside it are all background: the FAST kernel crosses
    // that region with one exact free-path sample instead of delta-tracking through it (track_pool.inc: exterior_hop).
    D.has_exterior = 0;
    {
      const int bg = order[0];
      int lo[3] = {D.brick_n[0], D.brick_n[1], D.brick_n[2]}, hi[3] = {-1, -1, -1};
      for (int bz = 0; bz < D.brick_n[2]; ++bz)
        for (int by = 0; by < D.brick_n[1]; ++by)
          for (int bx = 0; bx < D.brick_n[0]; ++bx) {
            const int b = (bz * D.brick_n[1] + by) * D.brick_n[0] + bx;
            if (!mixed[b] && first[b] == bg) continue;
            const int c3[3] = {bx, by, bz};
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c3[a]); hi[a] = std::max(hi[a], c3[a]); }
          }
      if (homogeneous[bg] > 0 && hi[0] >= 0 && !D.knobs.no_exterior) {
        long outside = 0;
        for (int b = 0; b < D.brick_count; ++b) {
          const int bx = b % D.brick_n[0], by = (b / D.brick_n[0]) % D.brick_n[1], bz = b / (D.brick_n[0] * D.brick_n[1]);
          const bool out = bx < lo[0] || bx > hi[0] || by < lo[1] || by > hi[1] || bz < lo[2] || bz > hi[2];
          if (out) { first[b] = -2; ++outside; }  // marks EXTERIOR for the encoder below
        }
        if (outside > 0) {
          D.has_exterior = 1;
          D.brick_palette[14] = bg;
          const int nvx[3] = {nx, ny, nz};
          for (int a = 0; a < 3; ++a) {
            D.objbox_lo[a] = (float)(lo[a] << k) * H.voxels.voxel_size[a];
            D.objbox_hi[a] = (float)std::min((hi[a] + 1) << k, nvx[a]) * H.voxels.voxel_size[a];
          }
        }
      }
    }
    D.brick_bytes = (D.brick_count + 1) / 2;
    std::vector<unsigned char> bricks(D.brick_bytes, 0xFF);
    D.bricks_mixed = 0;
    D.bricks_exterior = 0;
    for (int b = 0; b < D.brick_count; ++b) {
      int code = 0xF;
      if (first[b] == -2) { code = 14; ++D.bricks_exterior; }
      else if (!mixed[b] && first[b] >= 0) code = code_of[first[b]];
      D.bricks_mixed += (code == 0xF);
      const int sh = (b & 1) * 4;
      bricks[b >> 1] = (unsigned char)((bricks[b >> 1] & ~(0xF << sh)) | (code << sh));
    }
    D.bricks = D.put(bricks);
    {
      // Second-level codes (same 4-bit coding, no EXTERIOR): a flight step that lands in a mixed brick asks this table,
      // which stays in L2 (0.5-1 MB), before it asks the volume (64-128 MiB: Infinity Cache / HBM).  On a body-filling
      // volume 78 % of the tissue voxels lie in mixed 16^3 bricks but only 24 % in mixed 4^3 sub-bricks, and the voxel
      // gathers of the flight step were what bound that workload (1.45 KB of fabric traffic per history at 5e9 histories/s).
      std::vector<unsigned char> sub((nsub + 1) / 2, 0xFF);
      D.sub_mixed = 0;
      // Round 2 (x-fastest rows): worth its dependent L2 round trip where most bricks a photon meets are mixed (thorax +24 %).
      // Round 3: the volume is stored in 4x4x4 TILES, one tile = one 64-byte sector = one sub-brick -- asking the volume
      // directly now costs one sector like asking this table, without the second dependent round trip, and the tile it
      // brings in serves the neighbouring voxels of later photons.  Measured on one box (tools/ab_second_level.sh): thorax 14.27
      // -> 13.68 ms, CIRS 6.52 -> 6.25, Catphan 4.17 -> 4.09 with the table OFF.  So it is off unless MCGPU_SUB_BRICKS=1
      // asks for it (kept: it halves the fabric traffic where that is what binds, and the tests hold both routes to the
      // same tallies).
      const char* knob = getenv("MCGPU_SUB_BRICKS");
      const bool off = knob ? atoi(knob) == 0 : true;
      for (size_t b = 0; b < nsub; ++b) {
        const int code = (!off && first2[b] >= 0 && first2[b] < 0x100) ? code_of[first2[b]] : 0xF;
        D.sub_mixed += (code == 0xF);
        const int sh = (int)(b & 1) * 4;
        sub[b >> 1] = (unsigned char)((sub[b >> 1] & ~(0xF << sh)) | (code << sh));
      }
      D.sub = off ? nullptr : D.put(sub);
    }
  } else {
    D.vol_kind = kVolU16;
    D.vol = D.put(idx16);
    D.vol_bytes = nvox * 2;
    D.palette_size = (int)index_of.size();
    D.palette = D.put(palette);
  }

  // ---- cross-section records
  std::vector<float> wood(2 * (size_t)nv), rec(8 * (size_t)nv * nmat, 0.f);
  for (int i = 0; i < nv; ++i) { wood[2 * i] = H.mat.woodcock[i].x; wood[2 * i + 1] = H.mat.woodcock[i].y; }
  for (int i = 0; i < nv; ++i)
    for (int m = 0; m < kMaxMaterials; ++m) {
      const int mc = D.compact_of[m];
      if (mc < 0) continue;
      float* r = &rec[8 * ((size_t)mc * nv + i)];  // material-major rows (track_common.inc: table_row)
      const Float3& a = H.mat.a[(size_t)i * kMaxMaterials + m];
      const Float3& b = H.mat.b[(size_t)i * kMaxMaterials + m];
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = b.x; r[4] = b.y; r[5] = b.z;
      r[6] = H.mat.pmax[(size_t)(i + 1) * kMaxMaterials + m];
      r[7] = 0.f;
    }
  D.woodcock = D.put(wood);
  D.mfp = D.put(rec);
  {
    std::vector<float> tot(2 * (size_t)nv * nmat);
    for (size_t k = 0; k < (size_t)nv * nmat; ++k) { tot[2 * k] = rec[8 * k]; tot[2 * k + 1] = rec[8 * k + 3]; }
    D.mfp_tot = D.put(tot);
    // Brackets of the total cross section for the FAST flight step: per (coarse energy bin = 2^shift table bins,
    // material) the centre of [min, max] of a_tot + b_tot * E over the coarse bin as fp16, and per coarse bin one
    // relative half width covering every material.  A step whose random number falls outside
    // [1 - m*hi, 1 - m*lo) is decided from LDS alone; only the narrow band in between fetches the exact value, so the
    // decisions are those of the exact test.  The LDS image takes the finest table that still leaves two workgroups per CU.
    D.sig_tot_host = tot;
  }
  std::vector<float> xco(kRayleighPoints * nmat), pco(xco), aco(xco), bco(xco);
  std::vector<unsigned char> itl(kRayleighPoints * nmat), itu(itl);
  std::vector<float> fco(kMaxShells * nmat, 0.f), uico(fco), fj0(fco);
  std::vector<int> nosc(std::max(nmat, 1), 0);
  std::vector<float> shell_cut(kMaxShells * std::max(nmat, 1), 1.0f);
  std::vector<unsigned char> shell_alias(kMaxShells * std::max(nmat, 1), 0);
  for (int m = 0; m < kMaxMaterials; ++m) {
    const int mc = D.compact_of[m];
    if (mc < 0) continue;
    for (int i = 0; i < kRayleighPoints; ++i) {
      xco[mc * kRayleighPoints + i] = H.mat.xco[m * kRayleighPoints + i];
      pco[mc * kRayleighPoints + i] = H.mat.pco[m * kRayleighPoints + i];
      aco[mc * kRayleighPoints + i] = H.mat.aco[m * kRayleighPoints + i];
      bco[mc * kRayleighPoints + i] = H.mat.bco[m * kRayleighPoints + i];
      itl[mc * kRayleighPoints + i] = H.mat.itlco[m * kRayleighPoints + i];
      itu[mc * kRayleighPoints + i] = H.mat.ituco[m * kRayleighPoints + i];
    }
    for (int s = 0; s < kMaxShells; ++s) {
      fco[s * nmat + mc] = H.mat.fco[m + s * kMaxMaterials];
      uico[s * nmat + mc] = H.mat.uico[m + s * kMaxMaterials];
      fj0[s * nmat + mc] = H.mat.fj0[m + s * kMaxMaterials];
    }
    nosc[mc] = H.mat.noscco[m];
    {
      // FAST Compton sampler (track_common.inc: compton_draw): Walker alias table of the shell weights f_i (Vose's
      // construction, in double): column k keeps shell k below cut[k] and maps the rest of the column to alias[k]
      const int n = std::min(H.mat.noscco[m], kMaxShells);
      double F = 0.0;
      for (int s = 0; s < n; ++s) F += (double)H.mat.fco[m + s * kMaxMaterials];
      std::vector<double> q(n);
      std::vector<int> small, large;
      for (int s = 0; s < n; ++s) {
        q[s] = F > 0.0 ? (double)H.mat.fco[m + s * kMaxMaterials] * n / F : 1.0;
        (q[s] < 1.0 ? small : large).push_back(s);
        shell_cut[s * nmat + mc] = 1.0f;
        shell_alias[s * nmat + mc] = (unsigned char)s;
      }
      while (!small.empty() && !large.empty()) {
        const int a = small.back(), b = large.back();
        small.pop_back();
        shell_cut[a * nmat + mc] = (float)q[a];
        shell_alias[a * nmat + mc] = (unsigned char)b;
        q[b] -= 1.0 - q[a];
        if (q[b] < 1.0) { large.pop_back(); small.push_back(b); }
      }
    }
  }
  D.xco = D.put(xco); D.pco = D.put(pco); D.aco = D.put(aco); D.bco = D.put(bco);
  D.itl = D.put(itl); D.itu = D.put(itu);
  D.fco = D.put(fco); D.uico = D.put(uico); D.fj0 = D.put(fj0);
  D.s0_bounds = D.put(build_s0_bounds(H, D.compact_of, nmat, &D.s0_emin, &D.s0_inv_w));  // COMPAT: bounds of S0 per (material, energy bin)
  D.noscco = D.put(nosc);
  D.shell_cut = D.put(shell_cut);
  D.shell_alias = D.put(shell_alias);
  D.espc = D.put(std::vector<float>(H.spectrum.espc, H.spectrum.espc + kMaxSpectrumBins));
  D.cutoff = D.put(std::vector<float>(H.spectrum.cutoff, H.spectrum.cutoff + kMaxSpectrumBins));
  D.alias = D.put(std::vector<short>(H.spectrum.alias, H.spectrum.alias + kMaxSpectrumBins));
  // ---- LDS image of the kernels (byte offsets; track_common.inc: stage_tables)
  {
    LdsLayout& Y = D.lds;
    int off = 0;
    auto take = [&](int bytes, int align) { off = (off + align - 1) / align * align; const int at = off; off += bytes; return at; };
    const int ns = std::min(H.spectrum.num_bins + 1, kMaxSpectrumBins) + 1;
    int total_shells = 0;
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] >= 0) { D.shell_first[D.compact_of[m]] = total_shells; total_shells += std::min(H.mat.noscco[m], kMaxShells); }
    Y.shells = take(std::max(total_shells, 1) * 16, 16);
    Y.nosc = take(std::max(nmat, 1) * 8, 16);
    Y.espc = take(ns * 4, 16);
    Y.cutoff = take(ns * 4, 16);
    Y.alias = take(ns * 2, 16);
    Y.pal = take(D.vol_kind == kVolU8 ? (16 + D.palette_size) * 8 : 0, 16);
    Y.brick = take(D.vol_kind == kVolU8 ? D.brick_bytes : 0, 16);
    Y.dose_mat = take(2 * kMaxMaterials * 8, 16);
    Y.slots = take(0, 16);  // the COMPAT kernel's image ends here
    take(kSlotWords * kPoolParked * kPoolBlockThreads * 4, 16);
    Y.sig_mid = Y.sig_w = off;
    D.sig_shift = -1;
    if (!getenv("MCGPU_NO_BRACKETS") && nmat > 0) {
      const int budget = 160 * 1024 / 2;  // two 1024-thread workgroups per CU
      for (int shift = 6; shift <= 12; ++shift) {
        const int nc = (nv + (1 << shift) - 1) >> shift;
        const int need = (off + 15) / 16 * 16 + (nc * nmat * 2 + 15) / 16 * 16 + (nc * 4 + 15) / 16 * 16;
        if (need > budget) continue;
        D.sig_shift = shift;
        D.sig_coarse = nc;
        Y.sig_mid = take(nc * nmat * 2, 16);
        Y.sig_w = take(nc * 4, 16);
        break;
      }
    }
    Y.total = (off + 15) / 16 * 16;
  }
  if (D.sig_shift >= 0) {
    const int nc = D.sig_coarse, shift = D.sig_shift;
    auto to_half = [](float f) -> unsigned short {  // round to nearest even; inputs are positive normal numbers
      uint32_t x; memcpy(&x, &f, 4);
      const int e = (int)((x >> 23) & 0xFF) - 127 + 15;
      uint32_t m = x & 0x7FFFFF;
      if (e <= 0) return 0;
      if (e >= 31) return 0x7BFF;
      uint32_t h = ((uint32_t)e << 10) | (m >> 13);
      const uint32_t rem = m & 0x1FFF;
      if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) ++h;
      return (unsigned short)std::min<uint32_t>(h, 0x7BFF);
    };
    auto from_half = [](unsigned short h) -> double { return std::ldexp((double)((h & 0x3FF) | 0x400), (int)(h >> 10) - 25); };
    std::vector<unsigned short> mid((size_t)nc * nmat, 0);
    std::vector<float> wv(nc, 0.f);
    const double e0 = H.mat.e0, ide = H.mat.ide;
    for (int c = 0; c < nc; ++c) {
      double wmax = 0.0;
      for (int mc = 0; mc < nmat; ++mc) {
        double lo = 1e300, hi = -1e300;
        for (int i = c << shift; i < std::min(nv, (c + 1) << shift); ++i) {
          const double a = D.sig_tot_host[2 * ((size_t)mc * nv + i)], b = D.sig_tot_host[2 * ((size_t)mc * nv + i) + 1];
          // the kernel evaluates a + b * E for E in [E_i, E_{i+1}) (one table bin; a little beyond for float rounding)
          const double ea = e0 + (i - 0.01) / ide, eb = e0 + (i + 1.01) / ide;
          lo = std::min(lo, std::min(a + b * ea, a + b * eb));
          hi = std::max(hi, std::max(a + b * ea, a + b * eb));
        }
        if (!(lo > 0.0)) lo = std::min(1e-30, hi > 0.0 ? hi : 1e-30);
        const unsigned short hbits = to_half((float)(0.5 * (lo + hi)));
        mid[(size_t)c * nmat + mc] = hbits;
        const double m = from_half(hbits);
        if (hbits == 0 || hbits == 0x7BFF || !(m > 0.0)) { wmax = 1e30; continue; }  // not representable: the band is everything
        wmax = std::max(wmax, std::max((hi - m) / m, (m - lo) / m));
      }
      wv[c] = (float)std::min(wmax * 1.001 + 1e-5, 1e30);
    }
    D.sig_mid = D.put(mid);
    D.sig_w = D.put(wv);
  }
  D.num_spectrum_bins = H.spectrum.num_bins;
  {
    TrackCold cold;
    memset(&cold, 0, sizeof cold);
    cold.xco = D.xco; cold.pco = D.pco; cold.aco = D.aco; cold.bco = D.bco; cold.itl = D.itl; cold.itu = D.itu;
    cold.fco = D.fco; cold.uico = D.uico; cold.fj0 = D.fj0; cold.noscco = D.noscco;
    cold.s0_bounds = D.s0_bounds; cold.s0_emin = D.s0_emin; cold.s0_inv_w = D.s0_inv_w;
    cold.shell_cut = D.shell_cut; cold.shell_alias = D.shell_alias;
    cold.espc = D.espc; cold.cutoff = D.cutoff; cold.alias = D.alias;
    cold.bricks = D.bricks;
    cold.sig_mid = D.sig_mid; cold.sig_w = D.sig_w;
    for (int c = 0; c < 16; ++c) cold.brick_palette[c] = D.brick_palette[c];
    // dose tallies (read_input :1868-1893, init_CUDA_device :2636-2657,2694-2720)
    const SimConfig& cfg = H.cfg;
    if (cfg.flag_material_dose == 1) {
      D.dose_materials = D.put(std::vector<unsigned long long>(2 * kMaxMaterials, 0ULL));
      D.dose_flags |= kDoseMaterials;
    }
    if (cfg.dose_roi[1] > -1) {
      D.dose_roi_voxels = (size_t)(cfg.dose_roi[1] - cfg.dose_roi[0] + 1) * (size_t)(cfg.dose_roi[3] - cfg.dose_roi[2] + 1) *
                          (size_t)(cfg.dose_roi[5] - cfg.dose_roi[4] + 1);
      D.dose_voxels = D.put(std::vector<unsigned long long>(2 * D.dose_roi_voxels, 0ULL));
      D.dose_flags |= kDoseVoxels;
    }
    cold.dose_voxels = D.dose_voxels;
    cold.dose_materials = D.dose_materials;
    for (int k = 0; k < 6; ++k) cold.dose_roi[k] = cfg.dose_roi[k];
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] >= 0) cold.material_of_compact[D.compact_of[m]] = m;
    for (int m = 0; m < kMaxMaterials; ++m) cold.shell_first[m] = D.shell_first[m];
    for (int k = 0; k < 3; ++k) { cold.objbox_lo[k] = D.objbox_lo[k]; cold.objbox_hi[k] = D.objbox_hi[k]; }
    cold.thresh_compton = cold.thresh_rayleigh = cold.thresh_new = cold.flyable_low = cold.swap_batch = cold.trade_slots = -1;  // apply_schedule
    {
      // azimuthal aperture of the beam (the same for every projection: the pose rotates the beam frame, MC-GPU_v1.3.cu:3280-3434)
      cold.fan_ratio_lo = -3.0e38f;
      cold.fan_ratio_hi = 3.0e38f;
      bool same = !H.source.empty();
      for (const SourcePose& sp : H.source) same = same && sp.phi_low == H.source[0].phi_low && sp.D_phi == H.source[0].D_phi;
      if (same) {
        const double lo = (double)H.source[0].phi_low, hi = lo + (double)H.source[0].D_phi;
        if (lo > 1.0e-3 && hi < 3.14159265358979323846 - 1.0e-3 && hi > lo) {
          const double r_hi = std::cos(lo) / std::sin(lo), r_lo = std::cos(hi) / std::sin(hi);  // cot decreases on (0, pi)
          const double margin = 2.0e-6 * (r_hi - r_lo);
          cold.fan_ratio_lo = (float)(r_lo + margin);
          cold.fan_ratio_hi = (float)(r_hi - margin);
        }
      }
    }
    D.cold_host = cold;
    D.cold = D.put(std::vector<TrackCold>(1, cold));
    D.src_all = D.put(H.source);
    D.det_all = D.put(H.detector);
  }
  D.work_counter = D.put(std::vector<unsigned long long>((size_t)kNumCounters * kCounterStride, 0ULL));
  HIP_TRY(hipEventCreate(&D.ev_start));
  HIP_TRY(hipEventCreate(&D.ev_stop));
  apply_schedule(D);
  HIP_TRY(hipDeviceSynchronize());
}

void require(bool ok, int code, const char* msg) { if (!ok) throw Error(code, msg); }

TrackArgs make_args(const mcgpu_ctx& C, int p) {
  const HostModel& H = C.host;
  const DeviceModel& D = C.dev;
  TrackArgs A;
  memset(&A, 0, sizeof A);
  A.vol = D.vol; A.palette = D.palette; A.vol_kind = D.vol_kind; A.palette_size = D.palette_size;
  A.brick_shift = D.brick_shift; A.brick_nx = D.brick_n[0]; A.brick_nxy = D.brick_n[0] * D.brick_n[1];
  A.brick_bytes = D.vol_kind == kVolU8 ? D.brick_bytes : 0;
  A.sub = D.vol_kind == kVolU8 ? D.sub : nullptr; A.sub_nx = D.sub_n[0]; A.sub_nxy = D.sub_n[0] * D.sub_n[1];
  A.lds = D.lds;
  A.nx = H.voxels.n[0]; A.ny = H.voxels.n[1]; A.nz = H.voxels.n[2]; A.nxy = A.nx * A.ny;
  for (int k = 0; k < 3; ++k) {
    A.inv_vs[k] = H.voxels.inv_voxel_size[k];
    A.bbox[k] = H.voxels.size_bbox[k];
    // upper clamp of the FAST kernel: bbox - EPS_SOURCE (MC-GPU_v1.3.h:87), lowered until it indexes the last voxel
    float hi = A.bbox[k] - 0.000015f;
    while ((int)(hi * A.inv_vs[k]) > H.voxels.n[k] - 1) hi = std::nextafter(hi, 0.0f);
    A.bbox_hi[k] = hi;
  }
  require((long long)A.nx * A.ny < (1LL << 24) && (long long)H.voxels.count() < (1LL << 31), -2,
          "!!ERROR!! voxel grid too large for the 32-bit voxel index of the kernel");
  A.e0 = H.mat.e0; A.ide = H.mat.ide; A.num_values = H.mat.num_values; A.nmat = D.nmat;
  A.woodcock = D.woodcock; A.mfp = D.mfp; A.mfp_tot = D.mfp_tot; A.sig_shift = D.sig_shift;
  A.cold = D.cold;
  A.nbins = H.spectrum.num_bins;
  A.src = D.src_all + p; A.det = D.det_all + p;
  A.stream_key = (unsigned)p;
  A.dose_flags = D.dose_flags;

  A.has_exterior = (D.vol_kind == kVolU8 && D.has_exterior) ? D.knobs.exterior_mode : 0;  // bit 0: hop during flight, bit 1: hop at the source
  // batching thresholds of the COMPAT kernel (lanes of a wave64 holding such a history in their registers or their parking slot)
  // The Compton batch of the COMPAT kernel walks every electron shell of the material several times in the reference's own
  // arithmetic: with tissue tables (29-40 shells) it is 60-73 % of the kernel and wants FULL batches -- threshold 40 of 64 lanes
  // instead of 20: thorax +39 %, CIRS +32 % (tools/compat_sweep.py) -- while the 4-12 shells of the Catphan's plastics prefer
  // photons back in flight early (40: -21 %).  Chosen from the mean shell count of the materials in use; tallies do not depend on it.
  // Second sweep: the tally/source batch is cheap and should not hold lanes back (24 -> 12..16 lanes), which in turn lets the
  // Compton batch wait for 48: thorax 1.9e8 -> 2.95e8, CIRS 4.6e8 -> 6.5e8, Catphan 1.58e9 -> 1.66e9 histories/s.
  // Round 3, with resumable Compton trials and two batches per lane (track_kernel.inc), the same thresholds are still the best of
  // the sweep (profiles/r03u_compat_sweep.txt): thorax 4.1e8, CIRS 9.2e8, Catphan 1.8e9; exchanging the two histories of a lane
  // pays from 1-8 takers on (thresh_take), 16+ loses.
  // Later in round 3 the Compton batch lost two thirds of its cost (S0 bounds, in-place Klein-Nishina rejections) and the wave takes
  // four flight steps between two looks at its state: 48 / 8 / 16 on tissue and 32 / 4 / 24 on plastics sit on flat optima
  // (profiles/r03y_compat_sweep.txt).
  int shells = 0, used = 0;
  for (int m = 0; m < kMaxMaterials; ++m)
    if (D.compact_of[m] >= 0) { shells += std::min(H.mat.noscco[m], kMaxShells); ++used; }
  const bool many_shells = used > 0 && shells >= 20 * used;
  A.thresh_compton = D.knobs.compat_thresh[0] >= 0 ? D.knobs.compat_thresh[0] : (many_shells ? 48 : 32);
  A.thresh_rayleigh = D.knobs.compat_thresh[1] >= 0 ? D.knobs.compat_thresh[1] : (many_shells ? 8 : 4);
  A.thresh_new = D.knobs.compat_thresh[2] >= 0 ? D.knobs.compat_thresh[2] : (many_shells ? 16 : 24);
  A.thresh_take = D.knobs.compat_thresh[3] >= 0 ? D.knobs.compat_thresh[3] : 2;
  return A;
}


// After mcgpu_warp_geometry the voxels exist on the device only; whoever needs them on the host calls this first.
void sync_host_voxels(mcgpu_ctx& C) {
  if (!C.host_voxels_stale) return;
  HostModel& H = C.host;
  DeviceModel& D = C.dev;
  HIP_TRY(hipSetDevice(D.device_id));
  const size_t nvox = H.voxels.count();
  std::vector<unsigned char> idx(nvox);
  {  // the device volume is tiled (device_model.hpp: tiled_voxel)
    std::vector<unsigned char> tiled(D.vol_bytes);
    HIP_TRY(hipMemcpy(tiled.data(), D.vol, D.vol_bytes, hipMemcpyDeviceToHost));
    const int nx = H.voxels.n[0], ny = H.voxels.n[1], nz = H.voxels.n[2];
    const unsigned int snx = (unsigned int)D.sub_n[0], snxy = (unsigned int)(D.sub_n[0] * D.sub_n[1]);
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) idx[((size_t)z * ny + y) * nx + x] = tiled[tiled_voxel((unsigned)x, (unsigned)y, (unsigned)z, snx, snxy)];
  }
  int mat_of[256];
  float dens_of[256];
  for (int e = 0; e < D.palette_size; ++e) {
    int mc;
    memcpy(&mc, &D.palette_host[2 * e + 1], 4);
    int number = 1;
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] == mc) number = m + 1;
    mat_of[e] = number;
    dens_of[e] = D.palette_host[2 * e];
  }
  for (size_t i = 0; i < nvox; ++i) { H.voxels.material[i] = (uint8_t)mat_of[idx[i]]; H.voxels.density[i] = dens_of[idx[i]]; }
  C.host_voxels_stale = false;
}

const void* host_table(mcgpu_ctx& C, const std::string& name, size_t& bytes) {
  HostModel& H = C.host;
  if (name == "voxel_mat_dens") sync_host_voxels(C);
  auto cache = [&](const void* p, size_t n) -> const void* {
    auto& v = C.table_cache[name];
    v.assign((const unsigned char*)p, (const unsigned char*)p + n);
    bytes = n;
    return v.data();
  };
#define DIRECT(vec) do { bytes = (vec).size() * sizeof((vec)[0]); return (const void*)(vec).data(); } while (0)
  if (name == "source_data") DIRECT(H.source);
  if (name == "detector_data") DIRECT(H.detector);
  if (name == "mfp_woodcock") DIRECT(H.mat.woodcock);
  if (name == "mfp_a") DIRECT(H.mat.a);
  if (name == "mfp_b") DIRECT(H.mat.b);
  if (name == "xco") DIRECT(H.mat.xco);
  if (name == "pco") DIRECT(H.mat.pco);
  if (name == "aco") DIRECT(H.mat.aco);
  if (name == "bco") DIRECT(H.mat.bco);
  if (name == "pmax") DIRECT(H.mat.pmax);
  if (name == "itlco") DIRECT(H.mat.itlco);
  if (name == "ituco") DIRECT(H.mat.ituco);
  if (name == "fco") DIRECT(H.mat.fco);
  if (name == "uico") DIRECT(H.mat.uico);
  if (name == "fj0") DIRECT(H.mat.fj0);
#undef DIRECT
  if (name == "s0_bounds") {  // COMPAT kernel: {lo, hi} of S0 per (material number - 1, energy bin), then {emin, 1 / bin width}
    float emin = 0.f, inv_w = 0.f;
    std::vector<float> b = build_s0_bounds(H, nullptr, kMaxMaterials, &emin, &inv_w);
    b.push_back(emin);
    b.push_back(inv_w);
    return cache(b.data(), b.size() * sizeof(float));
  }
  if (name == "noscco") return cache(H.mat.noscco, sizeof H.mat.noscco);
  if (name == "espc") return cache(H.spectrum.espc, sizeof H.spectrum.espc);
  if (name == "espc_cutoff") return cache(H.spectrum.cutoff, sizeof H.spectrum.cutoff);
  if (name == "espc_alias") return cache(H.spectrum.alias, sizeof H.spectrum.alias);
  if (name == "density_max") return cache(H.voxels.density_max, sizeof H.voxels.density_max);
  if (name == "density_nominal") return cache(H.mat.density_nominal, sizeof H.mat.density_nominal);
  if (name == "voxel_size") return cache(H.voxels.voxel_size, sizeof H.voxels.voxel_size);
  if (name == "inv_voxel_size") return cache(H.voxels.inv_voxel_size, sizeof H.voxels.inv_voxel_size);
  if (name == "size_bbox") return cache(H.voxels.size_bbox, sizeof H.voxels.size_bbox);
  if (name == "voxel_mat_dens") {  // reference layout: float2 {material + 0.0001f, density} (MC-GPU_v1.3.cu:2135-2136)
    auto& v = C.table_cache[name];
    const size_t n = H.voxels.count();
    v.resize(n * 8);
    float* f = (float*)v.data();
    for (size_t i = 0; i < n; ++i) { f[2 * i] = (float)(H.voxels.material[i]) + 0.0001f; f[2 * i + 1] = H.voxels.density[i]; }
    bytes = v.size();
    return v.data();
  }
  return nullptr;
}

}  // namespace
}  // namespace mcgpu

using namespace mcgpu;

#define ABI_BEGIN try {
#define ABI_END                                               \
  }                                                           \
  catch (const Error& e) { return set_error(e.code, e.what()); } \
  catch (const std::exception& e) { return set_error(-2, e.what()); } \
  catch (...) { return set_error(-2, "unknown failure"); }

extern "C" {

int mcgpu_abi_version(void) { return 1; }
void mcgpu_set_last_error_(const char* message) { (void)set_error(-1, message ? message : "unknown failure"); }  // for scan.cpp
const char* mcgpu_last_error(void) { return g_last_error.c_str(); }

int mcgpu_create(const char* input_path, int device_id, mcgpu_ctx** out) {
  ABI_BEGIN
  require(input_path && out, -1, "!!ERROR!! mcgpu_create: null argument");
  std::unique_ptr<mcgpu_ctx> c(new mcgpu_ctx);
  load_model(input_path, c->host);
  if (device_id >= 0) {
    upload_model(*c, device_id);
    c->has_device = true;
  }
  *out = c.release();
  return 0;
  ABI_END
}

int mcgpu_clone(const mcgpu_ctx* src, int device_id, mcgpu_ctx** out) {
  ABI_BEGIN
  require(src && out, -1, "!!ERROR!! mcgpu_clone: null argument");
  std::unique_ptr<mcgpu_ctx> c(new mcgpu_ctx);
  sync_host_voxels(*const_cast<mcgpu_ctx*>(src));
  c->host = src->host;
  if (device_id >= 0) {
    upload_model(*c, device_id);
    c->has_device = true;
  }
  *out = c.release();
  return 0;
  ABI_END
}

void mcgpu_destroy(mcgpu_ctx* ctx) {
  if (!ctx) return;
  if (ctx->has_device) {
    (void)hipSetDevice(ctx->dev.device_id);
    ctx->dev.release();
  }
  delete ctx;
}

int mcgpu_config_i64(const mcgpu_ctx* ctx, const char* key, long long* value) {
  ABI_BEGIN
  require(ctx && key && value, -1, "!!ERROR!! mcgpu_config_i64: null argument");
  const SimConfig& c = ctx->host.cfg;
  const std::string k(key);
  if (k == "total_histories") *value = (long long)c.total_histories;
  else if (k == "seed") *value = c.seed;
  else if (k == "gpu_id") *value = c.gpu_id;
  else if (k == "threads_per_block") *value = c.threads_per_block;
  else if (k == "histories_per_thread") *value = c.histories_per_thread;
  else if (k == "num_projections") *value = c.num_projections;
  else if (k == "enable_specific_angles") *value = c.enable_specific_angles;
  else if (k == "flag_material_dose") *value = c.flag_material_dose;
  else if (k == "num_voxels_x") *value = ctx->host.voxels.n[0];
  else if (k == "num_voxels_y") *value = ctx->host.voxels.n[1];
  else if (k == "num_voxels_z") *value = ctx->host.voxels.n[2];
  else if (k == "num_pixels_x") *value = ctx->host.detector[0].nx;
  else if (k == "num_pixels_z") *value = ctx->host.detector[0].nz;
  else if (k == "num_energy_values") *value = ctx->host.mat.num_values;
  else if (k == "num_spectrum_bins") *value = ctx->host.spectrum.num_bins;
  else if (k == "num_materials_used") { int n = 0; for (int m = 0; m < kMaxMaterials; ++m) n += ctx->host.mat.used[m]; *value = n; }
  else if (k == "palette_size") *value = ctx->dev.palette_size;
  else if (k == "volume_kind") *value = ctx->dev.vol_kind;
  else if (k == "volume_bytes_device") *value = (long long)ctx->dev.vol_bytes;
  else if (k == "num_cus") *value = ctx->dev.num_cus;
  else if (k == "device_id") *value = ctx->has_device ? ctx->dev.device_id : -1;
  else if (k == "brick_shift") *value = ctx->dev.brick_shift;
  else if (k == "brick_count") *value = ctx->dev.brick_count;
  else if (k == "bricks_mixed") *value = ctx->dev.bricks_mixed;
  else if (k == "bricks_exterior") *value = ctx->dev.bricks_exterior;
  else if (k == "sub_bricks") *value = (long long)ctx->dev.sub_n[0] * ctx->dev.sub_n[1] * ctx->dev.sub_n[2];
  else if (k == "sub_bricks_mixed") *value = ctx->dev.sub_mixed;
  else if (k == "blocks_per_cu") *value = ctx->dev.resident_fast;
  else if (k == "lds_bytes_fast") *value = ctx->dev.lds.total;
  else if (k == "sigma_bracket_shift") *value = ctx->dev.sig_shift;
  else if (k == "lds_bytes_compat") *value = ctx->dev.lds.slots + 14 * kTrackBlockThreads * 4;  // tables + one parked history per lane (track_kernel.inc)
  else return set_error(-2, std::string("unknown integer key: ") + key);
  return 0;
  ABI_END
}

int mcgpu_config_f64(const mcgpu_ctx* ctx, const char* key, double* value) {
  ABI_BEGIN
  require(ctx && key && value, -1, "!!ERROR!! mcgpu_config_f64: null argument");
  const SimConfig& c = ctx->host.cfg;
  const std::string k(key);
  if (k == "D_angle") *value = c.D_angle;
  else if (k == "initial_angle") *value = c.initial_angle;
  else if (k == "angularROI_0") *value = c.angularROI_0;
  else if (k == "angularROI_1") *value = c.angularROI_1;
  else if (k == "SRotAxisD") *value = c.SRotAxisD;
  else if (k == "vertical_translation") *value = c.vertical_translation;
  else if (k == "mean_energy_spectrum") *value = ctx->host.spectrum.mean_energy;
  else if (k == "e0") *value = ctx->host.mat.e0;
  else if (k == "pixel_size_x_mm") *value = 10.0 / (double)ctx->host.detector[0].inv_pixel_size_X;
  else if (k == "pixel_size_z_mm") *value = 10.0 / (double)ctx->host.detector[0].inv_pixel_size_Z;
  else if (k == "ide") *value = ctx->host.mat.ide;
  else return set_error(-2, std::string("unknown float key: ") + key);
  return 0;
  ABI_END
}

int mcgpu_host_table(mcgpu_ctx* ctx, const char* name, const void** data, size_t* bytes) {
  ABI_BEGIN
  require(ctx && name && data && bytes, -1, "!!ERROR!! mcgpu_host_table: null argument");
  size_t n = 0;
  const void* p = host_table(*ctx, name, n);
  if (!p) return set_error(-2, std::string("unknown table: ") + name);
  *data = p;
  *bytes = n;
  return 0;
  ABI_END
}

int mcgpu_projection_file_name(const mcgpu_ctx* ctx, int p, char* buf, size_t buf_bytes) {
  ABI_BEGIN
  require(ctx && buf && p >= 0 && p < ctx->host.cfg.num_projections, -1, "!!ERROR!! mcgpu_projection_file_name: bad argument");
  const std::string s = projection_file_name(ctx->host, p);
  require(s.size() + 1 <= buf_bytes, -2, "!!ERROR!! mcgpu_projection_file_name: buffer too small");
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
  ABI_END
}

int mcgpu_image_words(const mcgpu_ctx* ctx, size_t* words) {
  ABI_BEGIN
  require(ctx && words, -1, "!!ERROR!! mcgpu_image_words: null argument");
  *words = (size_t)4 * ctx->host.detector[0].total_pixels;
  return 0;
  ABI_END
}

int mcgpu_launch_shape(unsigned long long histories, int threads_per_block, int histories_per_thread, int* blocks, int* hpt_out,
                       unsigned long long* total_histories) {
  ABI_BEGIN
  require(threads_per_block > 0 && histories_per_thread > 0, -2, "!!ERROR!! mcgpu_launch_shape: bad argument");
  const LaunchShape L = reference_launch_shape(histories, threads_per_block, histories_per_thread);
  if (blocks) *blocks = L.blocks;
  if (hpt_out) *hpt_out = L.hpt;
  if (total_histories) *total_histories = L.total_histories;
  return 0;
  ABI_END
}

int mcgpu_advance_seed(int batch_number, unsigned long long total_histories, int seed) {
  return ranecu_advance_seed(batch_number, total_histories, seed);
}

int mcgpu_launch_projection(mcgpu_ctx* ctx, int p, int mode, int seed, unsigned long long first, unsigned long long count, int hpt,
                            void* image_dev, void* hip_stream) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_launch_projection: the context has no device (created with device_id < 0)");
  require(p >= 0 && p < ctx->host.cfg.num_projections, -1, "!!ERROR!! mcgpu_launch_projection: projection index out of range");
  require(image_dev != nullptr, -1, "!!ERROR!! mcgpu_launch_projection: null image buffer");
  require(mode == MCGPU_MODE_FAST || mode == MCGPU_MODE_COMPAT || mode == MCGPU_MODE_FAST_STATS, -1, "!!ERROR!! mcgpu_launch_projection: unknown mode");
  DeviceModel& D = ctx->dev;
  HIP_TRY(hipSetDevice(D.device_id));
  hipStream_t stream = (hipStream_t)hip_stream;
  TrackArgs A = make_args(*ctx, p);
  A.image = (unsigned long long*)image_dev;
  A.seed = seed; A.hpt = hpt; A.first = first; A.count = count;
  HIP_TRY(hipEventRecord(D.ev_start, stream));
  if (count > 0) {
    if (mode == MCGPU_MODE_COMPAT) {
      require(hpt > 0, -2, "!!ERROR!! mcgpu_launch_projection: histories per thread must be positive in COMPAT mode");
      require(seed > 0 && seed < 2147483399, -2, "!!ERROR!! mcgpu_launch_projection: RANECU seed out of range");
      const unsigned long long blocks = (count + kTrackBlockThreads - 1) / kTrackBlockThreads;
      require(blocks <= 0x7fffffffULL, -2, "!!ERROR!! mcgpu_launch_projection: too many batches");
      if (D.knobs.compat_stats) {  // diagnostic build only (-DMC_COMPAT_STATS): the kernel's wave-level counters
        if (!D.stats) D.stats = D.put(std::vector<unsigned long long>(kNumStats + 3 * kWaveTrace, 0ULL));
        A.stats = D.stats;
      }
      HIP_TRY(launch_track_compat(A, (int)blocks, stream));
    } else {
      // persistent grid: exactly the resident workgroups (an over-subscribed grid would run a second, thin round).  No
      // getenv and no synchronisation here: the scheduler's parameters were uploaded by apply_schedule.
      if (D.resident_fast <= 0) {
        D.resident_fast = D.knobs.blocks_per_cu > 0 ? D.knobs.blocks_per_cu : occupancy_track_fast(A);
        if (D.resident_fast <= 0) D.resident_fast = 1;
      }
      const unsigned long long want = (count + kPoolBlockThreads - 1) / kPoolBlockThreads;
      unsigned long long resident = (unsigned long long)D.num_cus * (unsigned long long)D.resident_fast;
      resident += resident * (unsigned long long)D.knobs.grid_spare_percent / 100ULL;  // spare workgroups: see the kernel's prologue
      HIP_TRY(hipMemsetAsync(D.work_counter, 0, (size_t)kNumCounters * kCounterStride * 8, stream));
      A.work_counter = D.work_counter;
      if (mode == MCGPU_MODE_FAST_STATS) {
#if defined(MC_WITH_STATS) && MC_WITH_STATS
        if (!D.stats) D.stats = D.put(std::vector<unsigned long long>(kNumStats + 3 * kWaveTrace, 0ULL));
        A.stats = D.stats;
        HIP_TRY(launch_track_stats(A, (int)std::min(want, resident), stream));
#else
        throw Error(-2, "!!ERROR!! mcgpu_launch_projection: MCGPU_MODE_FAST_STATS needs the diagnostic library (libmcgpu_amd_stats.so, MCGPU_AMD_LIB)");
#endif
      } else {
        HIP_TRY(launch_track_fast(A, (int)std::min(want, resident), stream));
      }
    }
  }
  HIP_TRY(hipEventRecord(D.ev_stop, stream));
  D.timed = true;
  return 0;
  ABI_END
}

int mcgpu_set_fast_schedule(mcgpu_ctx* ctx, int thresh_compton, int thresh_rayleigh, int thresh_new, int flyable_low, int swap_batch) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_set_fast_schedule: no device context");
  require(thresh_compton >= 1 && thresh_rayleigh >= 1 && thresh_new >= 1 && flyable_low >= 1 && swap_batch >= 1 && thresh_compton <= 64 &&
              thresh_rayleigh <= 64 && thresh_new <= 64 && flyable_low <= 64 && swap_batch <= 64,
          -2, "!!ERROR!! mcgpu_set_fast_schedule: thresholds are lane counts in 1..64");
  const int v[5] = {thresh_compton, thresh_rayleigh, thresh_new, flyable_low, swap_batch};
  for (int k = 0; k < 5; ++k) ctx->dev.sched[k] = v[k];
  apply_schedule(ctx->dev);
  return 0;
  ABI_END
}

int mcgpu_reload_env_knobs(mcgpu_ctx* ctx) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_reload_env_knobs: no device context");
  read_env_knobs(ctx->dev);
  ctx->dev.resident_fast = 0;  // MCGPU_BLOCKS_PER_CU may have changed: asked again at the next launch
  apply_schedule(ctx->dev);
  return 0;
  ABI_END
}

int mcgpu_scheduler_stats_ex(mcgpu_ctx* ctx, unsigned long long* out, int capacity, int reset) {
  ABI_BEGIN
  require(ctx && ctx->has_device && out && capacity > 0, -1, "!!ERROR!! mcgpu_scheduler_stats: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  const int n = std::min(capacity, kNumStats + 3 * kWaveTrace);  // counters, then the wave trace (device_model.hpp)
  for (int k = 0; k < capacity; ++k) out[k] = 0;
  if (!ctx->dev.stats) return 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ctx->dev.stats, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (reset) HIP_TRY(hipMemset(ctx->dev.stats, 0, (size_t)(kNumStats + 3 * kWaveTrace) * 8));
  return 0;
  ABI_END
}

int mcgpu_scheduler_stats(mcgpu_ctx* ctx, unsigned long long* out8, int reset) { return mcgpu_scheduler_stats_ex(ctx, out8, 8, reset); }

int mcgpu_last_kernel_ms(mcgpu_ctx* ctx, float* ms) {
  ABI_BEGIN
  require(ctx && ctx->has_device && ms, -1, "!!ERROR!! mcgpu_last_kernel_ms: bad argument");
  require(ctx->dev.timed, -1, "!!ERROR!! mcgpu_last_kernel_ms: nothing launched yet");
  HIP_TRY(hipEventSynchronize(ctx->dev.ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, ctx->dev.ev_start, ctx->dev.ev_stop));
  return 0;
  ABI_END
}

int mcgpu_clear_image(mcgpu_ctx* ctx, void* image_dev, void* hip_stream) {
  ABI_BEGIN
  require(ctx && ctx->has_device && image_dev, -1, "!!ERROR!! mcgpu_clear_image: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  HIP_TRY(hipMemsetAsync(image_dev, 0, (size_t)32 * ctx->host.detector[0].total_pixels, (hipStream_t)hip_stream));
  return 0;
  ABI_END
}

int mcgpu_copy_to_host(mcgpu_ctx* ctx, const void* src_dev, void* dst_host, size_t bytes, void* hip_stream) {
  ABI_BEGIN
  require(ctx && ctx->has_device && src_dev && dst_host, -1, "!!ERROR!! mcgpu_copy_to_host: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  return 0;
  ABI_END
}

int mcgpu_run_projection(mcgpu_ctx* ctx, int p, int mode, int seed, unsigned long long first, unsigned long long count, int hpt,
                         uint64_t* image_host, double* kernel_seconds, unsigned long long* histories_done) {
  ABI_BEGIN
  require(ctx && ctx->has_device && image_host, -1, "!!ERROR!! mcgpu_run_projection: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  const size_t bytes = (size_t)32 * ctx->host.detector[0].total_pixels;
  if (!ctx->dev.scratch_image) ctx->dev.scratch_image = (unsigned long long*)ctx->dev.put(std::vector<unsigned char>(bytes, 0));  // kept for the next call
  void* img = ctx->dev.scratch_image;
  int rc = 0;
  try {
    HIP_TRY(hipMemset(img, 0, bytes));
    rc = mcgpu_launch_projection(ctx, p, mode, seed, first, count, hpt, img, nullptr);
    if (rc == 0) {
      float ms = 0.f;
      rc = mcgpu_last_kernel_ms(ctx, &ms);
      if (kernel_seconds) *kernel_seconds = ms * 1e-3;
      HIP_TRY(hipMemcpy(image_host, img, bytes, hipMemcpyDeviceToHost));
      if (histories_done) *histories_done = (mode == MCGPU_MODE_COMPAT) ? count * (unsigned long long)hpt : count;
    }
  } catch (...) {
    throw;
  }
  return rc;
  ABI_END
}

int mcgpu_write_projection(mcgpu_ctx* ctx, int p, const uint64_t* image_host, unsigned long long total_histories, double seconds,
                           const char* file_name) {
  ABI_BEGIN
  require(ctx && image_host && p >= 0 && p < ctx->host.cfg.num_projections, -1, "!!ERROR!! mcgpu_write_projection: bad argument");
  require(total_histories > 0, -2, "!!ERROR!! mcgpu_write_projection: zero histories");
  const std::string name = file_name ? std::string(file_name) : projection_file_name(ctx->host, p);
  write_projection_ascii(ctx->host, p, image_host, total_histories, seconds, name);
  return 0;
  ABI_END
}

// Layout of an AsciiSlot's row block in 8-byte words: row_len[nz], row_off[nz + 1], row_arg[nz], row_sum[nz] (double),
// row_max[nz] (double), flags
static size_t ascii_row_words(int nz) { return (size_t)5 * nz + 2; }

int mcgpu_format_projection(mcgpu_ctx* ctx, const void* image_dev, unsigned long long total_histories, int slot, void* hip_stream) {
  ABI_BEGIN
  require(ctx && ctx->has_device && image_dev && slot >= 0 && slot < MCGPU_ASCII_SLOTS && total_histories > 0, -1, "!!ERROR!! mcgpu_format_projection: bad argument");
  DeviceModel& D = ctx->dev;
  HIP_TRY(hipSetDevice(D.device_id));
  const DetectorPose& d0 = ctx->host.detector[0];
  const int nx = d0.nx, nz = d0.nz;
  const size_t npix = (size_t)nx * nz, words = ascii_row_words(nz);
  DeviceModel::AsciiSlot& S = D.ascii[slot];
  if (!S.text_dev) {
    // room for 11 integer digits per number (values below 1e11 eV/cm^2 per history; a tally of 1e8 125-keV photons in one
    // 0.04 cm pixel would be 3e5): the formatter flags a projection that needs more
    D.ascii_capacity = npix * (4 * (11 + 9) + 4) + (size_t)nz + 64;
    void* t = nullptr;
    HIP_TRY(hipMalloc(&t, D.ascii_capacity));
    D.allocations.push_back(t);  // freed by release()
    S.text_dev = (char*)t;
    S.rows_dev = D.put(std::vector<unsigned long long>(words, 0ULL));
    HIP_TRY(hipHostMalloc((void**)&S.text_host, D.ascii_capacity, hipHostMallocNonCoherent));
    HIP_TRY(hipHostMalloc((void**)&S.rows_host, words * 8, hipHostMallocDefault));
    HIP_TRY(hipStreamCreateWithFlags(&S.copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
  }
  AsciiArgs a;
  a.image = (const unsigned long long*)image_dev;
  a.nx = nx; a.nz = nz; a.npix = npix;
  a.norm = projection_norm(ctx->host, total_histories);
  a.text = S.text_dev; a.capacity = D.ascii_capacity;
  a.row_len = S.rows_dev; a.row_off = S.rows_dev + nz; a.row_arg = (long long*)(S.rows_dev + 2 * nz + 1);
  a.row_sum = (double*)(S.rows_dev + 3 * nz + 1); a.row_max = (double*)(S.rows_dev + 4 * nz + 1);
  a.flags = (unsigned int*)(S.rows_dev + 5 * nz + 1);
  HIP_TRY(launch_ascii_format(a, (hipStream_t)hip_stream));
  HIP_TRY(hipMemcpyAsync(S.rows_host, S.rows_dev, words * 8, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
  HIP_TRY(hipEventRecord(S.ready, (hipStream_t)hip_stream));  // mcgpu_write_formatted_projection waits for it itself
  return 0;
  ABI_END
}

int mcgpu_write_formatted_projection(mcgpu_ctx* ctx, int p, int slot, unsigned long long total_histories, double seconds, const char* file_name) {
  ABI_BEGIN
  require(ctx && ctx->has_device && slot >= 0 && slot < MCGPU_ASCII_SLOTS && p >= 0 && p < ctx->host.cfg.num_projections && total_histories > 0, -1,
          "!!ERROR!! mcgpu_write_formatted_projection: bad argument");
  DeviceModel& D = ctx->dev;
  DeviceModel::AsciiSlot& S = D.ascii[slot];
  require(S.text_dev != nullptr, -1, "!!ERROR!! mcgpu_write_formatted_projection: nothing was formatted in this slot");
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipEventSynchronize(S.ready));  // the row block (coherent pinned memory) and the text on the device are complete
  const int nz = ctx->host.detector[0].nz;
  const unsigned long long* rows = S.rows_host;
  const unsigned int flags = (unsigned int)rows[5 * nz + 1];
  require(flags == 0u, -3, "!!ERROR!! projection values outside the range of the device formatter (>= 1e11 eV/cm^2 per history)");
  const size_t bytes = (size_t)rows[2 * nz];  // row_off[nz]
  HIP_TRY(hipMemcpyAsync(S.text_host, S.text_dev, bytes, hipMemcpyDeviceToHost, S.copy_stream));
  HIP_TRY(hipStreamSynchronize(S.copy_stream));
  // footer inputs: the rows in order (the first of equal maxima wins, MC-GPU_v1.3.cu:2893-2897)
  const long long* arg = (const long long*)(rows + 2 * nz + 1);
  const double* sum = (const double*)(rows + 3 * nz + 1);
  const double* mx = (const double*)(rows + 4 * nz + 1);
  double integral = 0.0, maximum = -100.0;
  long max_pixel = 0;
  for (int z = 0; z < nz; ++z) {
    integral += sum[z];
    if (mx[z] > maximum) { maximum = mx[z]; max_pixel = (long)arg[z]; }
  }
  const std::string name = file_name ? std::string(file_name) : projection_file_name(ctx->host, p);
  write_projection_preformatted(ctx->host, p, S.text_host, bytes, integral, maximum, max_pixel, total_histories, seconds, name);
  return 0;
  ABI_END
}

int mcgpu_dose_info(const mcgpu_ctx* ctx, int* flags, int roi6[6], size_t* roi_voxels) {
  ABI_BEGIN
  require(ctx != nullptr, -1, "!!ERROR!! mcgpu_dose_info: null context");
  const SimConfig& c = ctx->host.cfg;
  const bool vox = c.dose_roi[1] > -1;
  if (flags) *flags = (c.flag_material_dose == 1 ? 1 : 0) | (vox ? 2 : 0);
  if (roi6) for (int k = 0; k < 6; ++k) roi6[k] = c.dose_roi[k];
  if (roi_voxels)
    *roi_voxels = vox ? (size_t)(c.dose_roi[1] - c.dose_roi[0] + 1) * (size_t)(c.dose_roi[3] - c.dose_roi[2] + 1) * (size_t)(c.dose_roi[5] - c.dose_roi[4] + 1) : 0;
  return 0;
  ABI_END
}

int mcgpu_dose_read(mcgpu_ctx* ctx, uint64_t* voxels_out, uint64_t* materials_out) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_dose_read: the context has no device");
  DeviceModel& D = ctx->dev;
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipDeviceSynchronize());
  if (voxels_out) {
    require(D.dose_voxels != nullptr, -2, "!!ERROR!! mcgpu_dose_read: the voxel dose tally is disabled in the input file");
    HIP_TRY(hipMemcpy(voxels_out, D.dose_voxels, D.dose_roi_voxels * 16, hipMemcpyDeviceToHost));
  }
  if (materials_out) {
    require(D.dose_materials != nullptr, -2, "!!ERROR!! mcgpu_dose_read: the material dose tally is disabled in the input file");
    HIP_TRY(hipMemcpy(materials_out, D.dose_materials, (size_t)kMaxMaterials * 16, hipMemcpyDeviceToHost));
  }
  return 0;
  ABI_END
}

int mcgpu_dose_clear(mcgpu_ctx* ctx) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_dose_clear: the context has no device");
  DeviceModel& D = ctx->dev;
  HIP_TRY(hipSetDevice(D.device_id));
  if (D.dose_voxels) HIP_TRY(hipMemset(D.dose_voxels, 0, D.dose_roi_voxels * 16));
  if (D.dose_materials) HIP_TRY(hipMemset(D.dose_materials, 0, (size_t)kMaxMaterials * 16));
  return 0;
  ABI_END
}

int mcgpu_write_dose_report(mcgpu_ctx* ctx, const uint64_t* voxels, const uint64_t* materials, unsigned long long histories_per_projection,
                            double seconds, char* log, size_t log_bytes) {
  ABI_BEGIN
  require(ctx != nullptr && histories_per_projection > 0, -1, "!!ERROR!! mcgpu_write_dose_report: bad argument");
  sync_host_voxels(*ctx);  // voxel densities and material masses come from the host copy
  std::string text;
  if (voxels) {
    require(ctx->host.cfg.dose_roi[1] > -1, -2, "!!ERROR!! mcgpu_write_dose_report: the voxel dose tally is disabled in the input file");
    write_voxel_dose_report(ctx->host, voxels, histories_per_projection, seconds, text);
  }
  if (materials) {
    double mass[kMaxMaterials];
    material_masses(ctx->host, mass);
    format_materials_dose_report(ctx->host, materials, histories_per_projection, mass, text);
  }
  if (log && log_bytes > 0) {
    const size_t n = std::min(text.size(), log_bytes - 1);
    memcpy(log, text.data(), n);
    log[n] = '\0';
  } else {
    fputs(text.c_str(), stdout);
    fflush(stdout);
  }
  return 0;
  ABI_END
}

int mcgpu_finalize_projection(mcgpu_ctx* ctx, void* image_dev, unsigned long long total_histories, int crop_nx, void* planes_dev, int clear_image,
                              void* hip_stream) {
  ABI_BEGIN
  require(ctx && ctx->has_device && image_dev && planes_dev && total_histories > 0, -1, "!!ERROR!! mcgpu_finalize_projection: bad argument");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  const DetectorPose& d0 = ctx->host.detector[0];
  const int cx = (crop_nx > 0 && crop_nx < d0.nx) ? crop_nx : d0.nx;
  const double SCALE = 1.0 / 100.0f;
  const double norm = SCALE * d0.inv_pixel_size_X * d0.inv_pixel_size_Z / ((double)total_histories);  // MC-GPU_v1.3.cu:2860-2861
  HIP_TRY(launch_finalize((unsigned long long*)image_dev, d0.nx, d0.nz, cx, norm, (float*)planes_dev, clear_image, (hipStream_t)hip_stream));
  return 0;
  ABI_END
}

int mcgpu_finalize_projection_host(const mcgpu_ctx* ctx, const uint64_t* image_host, unsigned long long total_histories, int crop_nx,
                                   float* planes_host) {
  ABI_BEGIN
  require(ctx && image_host && planes_host && total_histories > 0, -1, "!!ERROR!! mcgpu_finalize_projection_host: bad argument");
  finalize_projection_host(ctx->host, image_host, total_histories, crop_nx, planes_host);
  return 0;
  ABI_END
}

int mcgpu_stack_create(const char* path, int nx, int ny, int nslices, double spacing_x, double spacing_y, mcgpu_stack** out) {
  ABI_BEGIN
  require(path && out && nx > 0 && ny > 0 && nslices > 0, -1, "!!ERROR!! mcgpu_stack_create: bad argument");
  *out = reinterpret_cast<mcgpu_stack*>(mha_create(path, nx, ny, nslices, spacing_x, spacing_y));
  return 0;
  ABI_END
}
int mcgpu_stack_append(mcgpu_stack* stack, const float* plane) {
  ABI_BEGIN
  require(stack && plane, -1, "!!ERROR!! mcgpu_stack_append: null argument");
  mha_append(reinterpret_cast<MhaStack*>(stack), plane);
  return 0;
  ABI_END
}
int mcgpu_stack_finish(mcgpu_stack* stack, int replace_zeros, float* replacement_value) {
  ABI_BEGIN
  require(stack != nullptr, -1, "!!ERROR!! mcgpu_stack_finish: null argument");
  const float v = mha_finish(reinterpret_cast<MhaStack*>(stack), replace_zeros != 0);
  if (replacement_value) *replacement_value = v;
  return 0;
  ABI_END
}
int mcgpu_stack_read(const char* path, int dims3[3], float* data, size_t capacity_elements) {
  ABI_BEGIN
  require(path && dims3, -1, "!!ERROR!! mcgpu_stack_read: null argument");
  std::vector<float> v;
  mha_read(path, dims3, v);
  if (data) {
    require(capacity_elements >= v.size(), -2, "!!ERROR!! mcgpu_stack_read: buffer too small");
    memcpy(data, v.data(), v.size() * 4);
  }
  return 0;
  ABI_END
}
int mcgpu_normalize_stack(const char* total_stack, const char* air_stack, double sigma_y, double sigma_x, const char* out_stack, double spacing_x,
                          double spacing_y) {
  ABI_BEGIN
  require(total_stack && air_stack && out_stack, -1, "!!ERROR!! mcgpu_normalize_stack: null argument");
  normalize_stack(total_stack, air_stack, sigma_y, sigma_x, out_stack, spacing_x, spacing_y);
  return 0;
  ABI_END
}

int mcgpu_stack_write_slice(mcgpu_stack* stack, int slice, const float* plane) {
  ABI_BEGIN
  require(stack && plane, -1, "!!ERROR!! mcgpu_stack_write_slice: null argument");
  mha_write_slice(reinterpret_cast<MhaStack*>(stack), slice, plane);
  return 0;
  ABI_END
}

// ---- 4-D support: one resident context, many (geometry, projection angle) jobs (cbctmc/mc/simulation.py:527-710)
int mcgpu_set_projection_angles(mcgpu_ctx* ctx, int n, const float* angles_deg) {
  ABI_BEGIN
  require(ctx && angles_deg && n >= 2 && n <= kMaxProjections, -1, "!!ERROR!! mcgpu_set_projection_angles: need 2..1024 angles");
  HostModel& H = ctx->host;
  require(H.cfg.num_projections >= 2, -2,
          "!!ERROR!! mcgpu_set_projection_angles: the input file must define a CT trajectory (more than one projection)");
  H.cfg.enable_specific_angles = 1;
  H.cfg.specific_angles.assign(angles_deg, angles_deg + n);
  H.cfg.num_projections = n;
  H.source.resize(1);    // pose 0 is the input file's (MC-GPU_v1.3.cu:3313); the others follow the angles
  H.detector.resize(1);
  build_ct_trajectory(H);
  if (ctx->has_device) {
    DeviceModel& D = ctx->dev;
    HIP_TRY(hipSetDevice(D.device_id));
    HIP_TRY(hipDeviceSynchronize());
    D.src_all = D.put(H.source);  // the old arrays stay allocated until the context is destroyed (180 B per projection)
    D.det_all = D.put(H.detector);
  }
  return 0;
  ABI_END
}

int mcgpu_set_geometry_arrays(mcgpu_ctx* ctx, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density) {
  ABI_BEGIN
  require(ctx && n && spacing_cm && material && density && n[0] > 0 && n[1] > 0 && n[2] > 0, -1, "!!ERROR!! mcgpu_set_geometry_arrays: bad argument");
  HostModel& H = ctx->host;
  VoxelGrid v;
  for (int k = 0; k < 3; ++k) {
    v.n[k] = n[k];
    v.voxel_size[k] = spacing_cm[k];
    v.size_bbox[k] = v.n[k] * v.voxel_size[k];
    v.inv_voxel_size[k] = 1.0f / v.voxel_size[k];
  }
  const size_t nvox = v.count();
  v.material.assign(material, material + nvox);
  v.density.resize(nvox);
  for (int k = 0; k < kMaxMaterials; ++k) v.density_max[k] = -999.0f;
  // densities as the voxel file would carry them ("%.6f", cbctmc/mc/voxel_data.pyx:25), so that handing arrays over
  // in-process gives the tables -- and therefore the tallies -- of the file-based flow
  std::unordered_map<uint32_t, float> q;
  for (size_t i = 0; i < nvox; ++i) {
    uint32_t b;
    memcpy(&b, &density[i], 4);
    auto it = q.find(b);
    float d;
    if (it != q.end()) d = it->second;
    else {
      char t[64];
      snprintf(t, sizeof t, "%.6f", (double)density[i]);
      d = strtof(t, nullptr);
      q.emplace(b, d);
    }
    const int mat = material[i];
    require(mat >= 1 && mat <= kMaxMaterials, -2, "!!ERROR load_voxels!! Voxel material number out of range!!");
    require(d >= 1.0e-9f, -2, "!!ERROR load_voxels!! Voxel density can not be 0 or negative");
    v.density[i] = d;
    if (d > v.density_max[mat - 1]) v.density_max[mat - 1] = d;
  }
  // Everything that can fail is built beside the live model and swapped in at the end: after an error return the context
  // is what it was before the call.  The Woodcock majorant and the set of loaded materials depend on the volume
  // (MC-GPU_v1.3.cu:2220-2233,2294-2296), so the material tables are rebuilt.
  MaterialTables mat;
  load_material_files(H.cfg.file_materials, v, mat);
  int roi[6];
  for (int k = 0; k < 6; ++k) roi[k] = H.cfg.dose_roi[k];
  if (roi[1] > -1)
    for (int ax = 0; ax < 3; ++ax) roi[2 * ax + 1] = std::min(roi[2 * ax + 1], v.n[ax] - 1);
  std::swap(H.voxels, v);
  std::swap(H.mat, mat);
  int roi_old[6];
  for (int k = 0; k < 6; ++k) { roi_old[k] = H.cfg.dose_roi[k]; H.cfg.dose_roi[k] = roi[k]; }
  if (ctx->has_device) {
    const int dev = ctx->dev.device_id;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipDeviceSynchronize());
    DeviceModel old = std::move(ctx->dev);  // stays allocated until the new model is up
    ctx->dev = DeviceModel();
    try {
      upload_model(*ctx, dev);
    } catch (...) {
      ctx->dev.release();
      ctx->dev = std::move(old);
      std::swap(H.voxels, v);
      std::swap(H.mat, mat);
      for (int k = 0; k < 6; ++k) H.cfg.dose_roi[k] = roi_old[k];
      throw;
    }
    for (int k = 0; k < 5; ++k) ctx->dev.sched[k] = old.sched[k];  // the tuned FAST schedule survives a geometry change
    apply_schedule(ctx->dev);
    old.release();  // NB: the dose tallies belong to a geometry and restart from zero with the new one
  }
  ctx->host_voxels_stale = false;
  ctx->table_cache.clear();
  return 0;
  ABI_END
}

int mcgpu_warp_geometry(mcgpu_ctx* ctx, const float* displacement, int frame, int default_material, float default_density) {
  ABI_BEGIN
  require(ctx && ctx->has_device && displacement && (frame == 0 || frame == 1), -1, "!!ERROR!! mcgpu_warp_geometry: bad argument (the context needs a device)");
  HostModel& H = ctx->host;
  DeviceModel& D = ctx->dev;
  require(D.vol_kind == kVolU8, -5, "!!ERROR!! mcgpu_warp_geometry: needs a palette volume (<= 256 distinct (material, density) pairs); use mcgpu_set_geometry_arrays");
  require(default_material >= 1 && default_material <= kMaxMaterials && D.compact_of[default_material - 1] >= 0, -5,
          "!!ERROR!! mcgpu_warp_geometry: the default material has no data file in this simulation");
  int default_index = -1;
  {
    char t[64];
    snprintf(t, sizeof t, "%.6f", (double)default_density);  // densities as a voxel file would carry them
    const float dq = strtof(t, nullptr);
    for (int e = 0; e < D.palette_size && default_index < 0; ++e) {
      int mc;
      memcpy(&mc, &D.palette_host[2 * e + 1], 4);
      if (mc == D.compact_of[default_material - 1] && D.palette_host[2 * e] == dq) default_index = e;
    }
  }
  require(default_index >= 0, -5, "!!ERROR!! mcgpu_warp_geometry: the default (material, density) is not in the palette; use mcgpu_set_geometry_arrays");
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipDeviceSynchronize());
  const size_t nvox = H.voxels.count();
  const size_t nsub = (size_t)D.sub_n[0] * D.sub_n[1] * D.sub_n[2];
  if (!D.vol_base) {  // first call: what is resident now is the base geometry of every later warp
    D.vol_base = D.put(std::vector<unsigned char>(D.vol_bytes, 0));
    HIP_TRY(hipMemcpy(D.vol_base, D.vol, D.vol_bytes, hipMemcpyDeviceToDevice));
    D.sub_first = D.put(std::vector<unsigned short>(nsub, 0));
    D.brick_first = D.put(std::vector<unsigned short>((size_t)D.brick_count, 0));
    D.code_of_dev = D.put(std::vector<unsigned char>(D.code_of, D.code_of + 256));
    D.rebuild_out = D.put(std::vector<unsigned int>(32, 0u));
    D.dvf = D.put(std::vector<float>(3 * nvox, 0.f));
  }
  HIP_TRY(hipMemcpy(D.dvf, displacement, 3 * nvox * 4, hipMemcpyHostToDevice));
  GeometryRebuild g;
  g.nx = H.voxels.n[0]; g.ny = H.voxels.n[1]; g.nz = H.voxels.n[2];
  g.brick_shift = D.brick_shift;
  for (int k = 0; k < 3; ++k) { g.bn[k] = D.brick_n[k]; g.sn[k] = D.sub_n[k]; }
  g.base_idx = D.vol_base; g.dvf = D.dvf; g.default_index = (unsigned char)default_index;
  g.idx = (unsigned char*)D.vol;
  g.sub_first = D.sub_first; g.brick_first = D.brick_first;
  g.sub = D.sub; g.bricks = D.bricks; g.code_of = D.code_of_dev; g.background = D.background;
  g.out = D.rebuild_out;
  const bool allow_exterior = !D.knobs.no_exterior;
  HIP_TRY(launch_geometry_rebuild(g, frame, allow_exterior, nullptr));
  unsigned int out[17];
  HIP_TRY(hipMemcpy(out, D.rebuild_out, sizeof out, hipMemcpyDeviceToHost));  // waits for the kernels
  // largest density per material among the palette entries that occur -> Woodcock majorant (the only table that depends on it)
  for (int m = 0; m < kMaxMaterials; ++m) H.voxels.density_max[m] = -999.0f;
  for (int e = 0; e < D.palette_size; ++e)
    if (out[e >> 5] & (1u << (e & 31))) {
      int mc;
      memcpy(&mc, &D.palette_host[2 * e + 1], 4);
      for (int m = 0; m < kMaxMaterials; ++m)
        if (D.compact_of[m] == mc) H.voxels.density_max[m] = std::max(H.voxels.density_max[m], D.palette_host[2 * e]);
    }
  rebuild_woodcock(H.mat, H.voxels.density_max);
  {
    std::vector<float> wood(2 * (size_t)H.mat.num_values);
    for (int i = 0; i < H.mat.num_values; ++i) { wood[2 * i] = H.mat.woodcock[i].x; wood[2 * i + 1] = H.mat.woodcock[i].y; }
    HIP_TRY(hipMemcpy(D.woodcock, wood.data(), wood.size() * 4, hipMemcpyHostToDevice));
  }
  D.bricks_mixed = (int)out[14]; D.bricks_exterior = (int)out[15]; D.sub_mixed = (int)out[16];
  const int had_exterior = D.has_exterior;
  D.has_exterior = out[15] > 0 ? 1 : 0;
  if (D.has_exterior) {
    const int k = D.brick_shift;
    for (int a = 0; a < 3; ++a) {
      D.objbox_lo[a] = (float)((int)out[8 + a] << k) * H.voxels.voxel_size[a];
      D.objbox_hi[a] = (float)std::min(((int)out[11 + a] + 1) << k, H.voxels.n[a]) * H.voxels.voxel_size[a];
      D.cold_host.objbox_lo[a] = D.objbox_lo[a];
      D.cold_host.objbox_hi[a] = D.objbox_hi[a];
    }
    // code 14 means "background outside the object box" (pack_codes_kernel): its palette slot must name the background
    // even when the BASE geometry had no exterior (its object box spanned the whole brick grid) and the warp made one
    D.brick_palette[14] = D.background;
    D.cold_host.brick_palette[14] = D.background;
  }
  if (D.has_exterior || had_exterior) HIP_TRY(hipMemcpy(D.cold, &D.cold_host, sizeof D.cold_host, hipMemcpyHostToDevice));
  ctx->host_voxels_stale = true;
  ctx->table_cache.clear();
  return 0;
  ABI_END
}

int mcgpu_warp_volume(mcgpu_ctx* ctx, const int n[3], const uint8_t* material, const float* density, const float* displacement,
                      int default_material, float default_density, uint8_t* material_out, float* density_out) {
  ABI_BEGIN
  require(ctx && ctx->has_device && n && material && density && displacement && material_out && density_out, -1,
          "!!ERROR!! mcgpu_warp_volume: bad argument (the context needs a device)");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  const size_t nvox = (size_t)n[0] * n[1] * n[2];
  unsigned char *m_in = nullptr, *m_out = nullptr;
  float *d_in = nullptr, *d_out = nullptr, *u = nullptr;
  hipError_t err = hipSuccess;
  auto step = [&](hipError_t e) { if (err == hipSuccess) err = e; };
  step(hipMalloc((void**)&m_in, nvox)); step(hipMalloc((void**)&m_out, nvox));
  step(hipMalloc((void**)&d_in, nvox * 4)); step(hipMalloc((void**)&d_out, nvox * 4)); step(hipMalloc((void**)&u, nvox * 12));
  if (err == hipSuccess) {
    step(hipMemcpy(m_in, material, nvox, hipMemcpyHostToDevice));
    step(hipMemcpy(d_in, density, nvox * 4, hipMemcpyHostToDevice));
    step(hipMemcpy(u, displacement, nvox * 12, hipMemcpyHostToDevice));
    if (err == hipSuccess) step(launch_warp(n[0], n[1], n[2], m_in, d_in, u, (unsigned char)default_material, default_density, m_out, d_out, nullptr));
    step(hipMemcpy(material_out, m_out, nvox, hipMemcpyDeviceToHost));
    step(hipMemcpy(density_out, d_out, nvox * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(m_in); (void)hipFree(m_out); (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(u);
  HIP_TRY(err);
  return 0;
  ABI_END
}

int mcgpu_write_voxel_file(const char* path, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density,
                           int gzip) {
  ABI_BEGIN
  require(path && n && spacing_cm && material && density, -1, "!!ERROR!! mcgpu_write_voxel_file: null argument");
  write_voxel_file(path, n, spacing_cm, material, density, gzip != 0);
  return 0;
  ABI_END
}

int mcgpu_write_voxel_binary(const char* path, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density) {
  ABI_BEGIN
  require(path && n && spacing_cm && material && density, -1, "!!ERROR!! mcgpu_write_voxel_binary: null argument");
  write_voxel_binary(path, n, spacing_cm, material, density);
  return 0;
  ABI_END
}

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

}  // extern "C"
