// engine.cpp -- the C ABI (include/mcgpu_amd.h) of an engine context: creation, configuration, launches, projection output,
// dose tallies, stacks.  Replaces the per-projection driver of main() (docker/mcgpu/MC-GPU_v1.3.cu:667-1056).  Compiled with hipcc.
#include "engine_internal.hpp"

using namespace mcgpu;

namespace mcgpu {
namespace {
thread_local std::string g_last_error;
}
int set_error(int code, const std::string& msg) {
  g_last_error = msg.find("ERROR") == std::string::npos ? "!!ERROR!! " + msg : msg;
  return code;
}
}  // namespace mcgpu

extern "C" {

int mcgpu_abi_version(void) { return 1; }
void mcgpu_set_last_error_(const char* message) { (void)set_error(-1, message ? message : "unknown failure"); }  // for scan.cpp
const char* mcgpu_last_error(void) { return g_last_error.c_str(); }

int mcgpu_create(const char* input_path, int device_id, mcgpu_ctx** out) {
  ABI_BEGIN
  require(input_path && out, -1, "!!ERROR!! mcgpu_create: null argument");
  knobs_warn_unknown();  // a misspelt MCGPU_* variable is reported, not silently ignored
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
  else if (k == "exterior_cylinder") *value = ctx->dev.ell_inv[0] > 0.f ? 1 : 0;  // object region = box AND elliptic cylinder (mark_exterior_region)
  else if (k == "sub_bricks") *value = (long long)ctx->dev.sub_n[0] * ctx->dev.sub_n[1] * ctx->dev.sub_n[2];
  else if (k == "sub_bricks_mixed") *value = ctx->dev.sub_mixed;
  else if (k == "tile_records") *value = ctx->dev.tile_rec ? 1 : 0;
  else if (k == "sub_brick_table") *value = ctx->dev.sub ? 1 : 0;
  else if (k == "segment_loop") *value = make_args(*ctx, 0).segment_loop;  // FAST per-wave kernel variant: flight segment as an inner loop (tissue volumes)
  else if (k == "fast_scheduler") *value = ctx->dev.knobs.fast_sched;  // 0: per-wave pools, 1: workgroup-level pool (MCGPU_FAST_SCHED at upload)
  else if (k == "tiles_in_mixed_bricks") *value = ctx->dev.tiles_in_mixed_bricks;
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
  require(mode == MCGPU_MODE_FAST || mode == MCGPU_MODE_COMPAT || mode == MCGPU_MODE_FAST_STATS || mode == MCGPU_MODE_FAST_F64, -1, "!!ERROR!! mcgpu_launch_projection: unknown mode");
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
      // environment read and no synchronisation here: the scheduler's parameters were uploaded by apply_schedule.
      if (D.resident_fast <= 0) {
        D.resident_fast = D.knobs.blocks_per_cu > 0 ? D.knobs.blocks_per_cu : std::min(occupancy_track_fast(A), occupancy_track_fast64(A));
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
      } else if (mode == MCGPU_MODE_FAST_F64) {
        HIP_TRY(launch_track_fast64(A, (int)std::min(want, resident), stream));
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
  ctx->dev.sched_set = true;
  apply_schedule(ctx->dev);
  return 0;
  ABI_END
}

int mcgpu_reload_env_knobs(mcgpu_ctx* ctx) {
  ABI_BEGIN
  require(ctx && ctx->has_device, -1, "!!ERROR!! mcgpu_reload_env_knobs: no device context");
  const int sched_kind = ctx->dev.knobs.fast_sched;  // fixed when the model was uploaded (it shapes the LDS image)
  read_env_knobs(ctx->dev);
  ctx->dev.knobs.fast_sched = sched_kind;
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

}  // extern "C"
