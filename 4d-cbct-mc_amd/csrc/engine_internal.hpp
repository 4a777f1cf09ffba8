#pragma once
// engine_internal.hpp -- what the translation units of the engine library share: the device model, the context, error plumbing.
// (engine.cpp: the C ABI of a context; model_device.cpp: device upload and launch arguments; engine_geometry.cpp: geometry changes of
// a resident context; engine_kat.cpp: known-answer and micro-benchmark hooks.)
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
#include "knobs.hpp"
#include "device_model.hpp"
#include "ascii_device.hpp"
#include "geometry_device.hpp"

namespace mcgpu {

hipError_t launch_track_compat(const TrackArgs& args, int blocks, hipStream_t stream);
hipError_t launch_track_fast(const TrackArgs& args, int blocks, hipStream_t stream);
int occupancy_track_fast(const TrackArgs& args);
hipError_t launch_track_fast64(const TrackArgs& args, int blocks, hipStream_t stream);  // MCGPU_MODE_FAST_F64 (track_fast64.hip)
int occupancy_track_fast64(const TrackArgs& args);
hipError_t launch_kat_fast64(int n, const unsigned int* u, const double* a, const double* b, const double* c, const float* dir, double* out, hipStream_t stream);
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


int set_error(int code, const std::string& msg);  // engine.cpp: records the calling thread's last error, returns `code`

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
  TileRecord* tile_rec = nullptr;  // second level as 16-byte records of the tiles (MCGPU_TILE_RECORDS; device_model.hpp), or null
  int rec_n[3] = {1, 1, 1};        // cubes of 2x2x2 tiles per axis
  long long tiles_in_mixed_bricks = 0;  // tiles a flight step can ask the second level / the volume for (hot set of the voxel gathers)
  int brick_shift = 0, brick_n[3] = {1, 1, 1}, brick_count = 0, brick_bytes = 0, bricks_mixed = 0;
  int brick_palette[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int has_exterior = 0, bricks_exterior = 0;
  float objbox_lo[3] = {0, 0, 0}, objbox_hi[3] = {0, 0, 0};
  float ell_c[2] = {0, 0}, ell_inv[2] = {0, 0};  // elliptic cylinder around the object (TrackCold::ell_*); inv 0: none
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
  float* wood_coarse = nullptr;  // FAST: majorant per coarse energy bin (LdsLayout::wood), rebuilt with the Woodcock table
  unsigned short* sig_mid = nullptr;  // cross-section brackets (FAST flight step), see upload_model
  float* sig_w = nullptr;
  int sig_shift = -1, sig_coarse = 0;
  int sched[5] = {40, 12, 44, 12, 40};  // FAST batching thresholds {compton, rayleigh, new, flyable_low, swap_batch} (mcgpu_set_fast_schedule; re-tuned in round 5: profiles/r05p_*)
  bool sched_set = false;              // mcgpu_set_fast_schedule has been called (else the scheduler's own defaults apply)
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
    int segment_loop = -1;                           // MCGPU_SEGMENT_LOOP: -1 chosen from the model (make_args), 0 / 1 forced
    int fast_sched = 0;                              // MCGPU_FAST_SCHED: 0 per-wave pools, 1 workgroup-level pool (fixes the LDS layout: read at upload)
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

}  // namespace mcgpu

struct mcgpu_ctx {
  mcgpu::HostModel host;
  mcgpu::DeviceModel dev;
  bool has_device = false;
  bool host_voxels_stale = false;  // the device holds a geometry warped there (mcgpu_warp_geometry): H.voxels is downloaded on demand
  std::map<std::string, std::vector<unsigned char>> table_cache;
};

namespace mcgpu {
// model_device.cpp
std::vector<float> coarse_woodcock(const HostModel& H);  // LdsLayout::wood from the host's Woodcock table
void mark_exterior_region(const HostModel& H, DeviceModel& D, const std::vector<unsigned char>& object, bool have_background,
                          std::vector<unsigned char>& exterior);  // object box + elliptic cylinder, exterior bricks (upload and device-side warp)
void read_env_knobs(DeviceModel& D);
void apply_schedule(DeviceModel& D);
void upload_model(mcgpu_ctx& C, int device_id);
void require(bool ok, int code, const char* msg);
TrackArgs make_args(const mcgpu_ctx& C, int p);
void sync_host_voxels(mcgpu_ctx& C);
const void* host_table(mcgpu_ctx& C, const std::string& name, size_t& bytes);
}  // namespace mcgpu


#define ABI_BEGIN try {
#define ABI_END                                               \
  }                                                           \
  catch (const Error& e) { return set_error(e.code, e.what()); } \
  catch (const std::exception& e) { return set_error(-2, e.what()); } \
  catch (...) { return set_error(-2, "unknown failure"); }

