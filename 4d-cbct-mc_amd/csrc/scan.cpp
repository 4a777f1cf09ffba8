// scan.cpp -- mcgpu_run_scan: the projection loop of main() (docker/mcgpu/MC-GPU_v1.3.cu:667-1056) as a device/host
// pipeline, written on top of the engine's own C ABI.
//
// Reference flow per projection, all on one host thread: kernel -> D2H of the 45 MB tally -> (MPI reduce) -> ~0.9 s of
// fprintf -> re-zero.  Here, per projection and on one HIP stream: track kernel -> [u64 copy to pinned memory, only when
// the ASCII files are wanted] -> finalize kernel (float32 planes + clears the tally) -> copy of the planes (9 MB) to
// one of two pinned buffers -> event.  A writer thread waits on the event and appends the planes to the three MetaImage
// stacks (and formats the ASCII file) while the GPU is already tracking the next projection.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcgpu_amd.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ScanError {
  int code;
  std::string msg;
};
#define HIP_OK(expr)                                                                                                \
  do {                                                                                                              \
    hipError_t _e = (expr);                                                                                         \
    if (_e != hipSuccess) throw ScanError{-1, std::string("!!HIP ERROR!! ") + #expr + ": " + hipGetErrorString(_e)}; \
  } while (0)
#define ABI_OK(expr)                                                    \
  do {                                                                  \
    const int _rc = (expr);                                             \
    if (_rc != 0) throw ScanError{_rc, std::string(mcgpu_last_error())}; \
  } while (0)

}  // namespace

extern "C" void mcgpu_set_last_error_(const char* message);  // engine.cpp (not part of the public ABI)

extern "C" int mcgpu_run_scan(mcgpu_ctx* ctx, const mcgpu_scan_options* opt, mcgpu_scan_report* report) {
  if (!ctx || !opt) { mcgpu_set_last_error_("!!ERROR!! mcgpu_run_scan: null argument"); return -1; }
  void *image_dev = nullptr, *planes_dev[2] = {nullptr, nullptr};
  float* planes_host[2] = {nullptr, nullptr};
  uint64_t* image_host[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  mcgpu_stack* stacks[3] = {nullptr, nullptr, nullptr};
  std::thread writer;
  struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    int queued = 0, written = 0;  // projections handed to / finished by the writer
    bool abort = false;
    std::string error;
    double writer_s = 0.0;
  } sh;
  int rc = 0;
  try {
    long long nproj_all = 1, hist_in = 0, seed = 0, tpb = 128, hpt = 150, nx = 0, nz = 0, dev = 0;
    ABI_OK(mcgpu_config_i64(ctx, "num_projections", &nproj_all));
    ABI_OK(mcgpu_config_i64(ctx, "total_histories", &hist_in));
    ABI_OK(mcgpu_config_i64(ctx, "seed", &seed));
    ABI_OK(mcgpu_config_i64(ctx, "threads_per_block", &tpb));
    ABI_OK(mcgpu_config_i64(ctx, "histories_per_thread", &hpt));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_x", &nx));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_z", &nz));
    ABI_OK(mcgpu_config_i64(ctx, "device_id", &dev));
    if (dev < 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: the context has no device"};
    double px_x = 0, px_z = 0;
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_x_mm", &px_x));
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_z_mm", &px_z));
    const int mode = opt->mode == MCGPU_MODE_COMPAT ? MCGPU_MODE_COMPAT : MCGPU_MODE_FAST;
    const int first = opt->first_projection > 0 ? opt->first_projection : 0;
    const int count = (opt->num_projections > 0) ? opt->num_projections : (int)nproj_all - first;
    if (first + count > nproj_all || count <= 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: projection range outside the trajectory"};
    unsigned long long H = opt->histories_per_projection ? opt->histories_per_projection : (unsigned long long)hist_in;
    // An input value below 95000 is a time budget in seconds per projection, not a history count (MC-GPU_v1.3.cu:650-655,
    // :689-809: the reference runs a speed test and converts).  Calibrate on the first projection with a throw-away launch.
    const bool by_time = !opt->histories_per_projection && hist_in < 95000;
    int blocks = 1, hpt_eff = (int)hpt;
    unsigned long long total = H;
    if (mode == MCGPU_MODE_COMPAT) ABI_OK(mcgpu_launch_shape(H, (int)tpb, (int)hpt, &blocks, &hpt_eff, &total));
    unsigned long long units = mode == MCGPU_MODE_COMPAT ? (unsigned long long)blocks * (unsigned long long)tpb : total;
    const int cx = (opt->crop_nx > 0 && opt->crop_nx < nx) ? opt->crop_nx : (int)nx;
    const size_t words = (size_t)4 * nx * nz, plane = (size_t)cx * nz;
    const double sx = opt->pixel_spacing_x > 0 ? opt->pixel_spacing_x : px_x, sy = opt->pixel_spacing_y > 0 ? opt->pixel_spacing_y : px_z;
    std::string folder;
    if (opt->output_folder) folder = opt->output_folder;
    else {
      char name[1024];
      ABI_OK(mcgpu_projection_file_name(ctx, 0, name, sizeof name));
      folder = name;
      const size_t slash = folder.find_last_of('/');
      folder = slash == std::string::npos ? "." : folder.substr(0, slash);
    }

    HIP_OK(hipSetDevice((int)dev));
    const unsigned int pinned_flags = getenv("MCGPU_PINNED_COHERENT") ? hipHostMallocDefault : hipHostMallocNonCoherent;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipMalloc(&image_dev, words * 8));
    HIP_OK(hipMemsetAsync(image_dev, 0, words * 8, stream));
    if (by_time) {
      const unsigned long long probe = 4000000ULL;
      const unsigned long long probe_units = mode == MCGPU_MODE_COMPAT ? (probe + (unsigned long long)hpt - 1) / (unsigned long long)hpt : probe;
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {  // the first launch pays one-off costs
        ABI_OK(mcgpu_launch_projection(ctx, first, mode, (int)seed, 0, probe_units, (int)hpt, image_dev, stream));
        ABI_OK(mcgpu_last_kernel_ms(ctx, &ms));
      }
      HIP_OK(hipMemsetAsync(image_dev, 0, words * 8, stream));
      ABI_OK(mcgpu_dose_clear(ctx));
      const double rate = (double)probe / (ms > 0.f ? ms * 1e-3 : 1e-3);
      H = (unsigned long long)(rate * (double)hist_in);
      if (H < 100000ULL) H = 100000ULL;
      total = H;
      if (mode == MCGPU_MODE_COMPAT) ABI_OK(mcgpu_launch_shape(H, (int)tpb, (int)hpt, &blocks, &hpt_eff, &total));
      units = mode == MCGPU_MODE_COMPAT ? (unsigned long long)blocks * (unsigned long long)tpb : total;
      if (opt->progress) {
        printf("       Time-limited run: %lld s per projection at %.3e x-rays/s -> %llu histories per projection\n", hist_in, rate, total);
        fflush(stdout);
      }
    }
    for (int b = 0; b < 2; ++b) {
      HIP_OK(hipMalloc(&planes_dev[b], 3 * plane * 4));
      // non-coherent (CPU-cacheable) pinned memory: the writer thread reads every byte; the event orders the accesses
      HIP_OK(hipHostMalloc((void**)&planes_host[b], 3 * plane * 4, pinned_flags));
      if (opt->write_ascii) HIP_OK(hipHostMalloc((void**)&image_host[b], words * 8, pinned_flags));
      HIP_OK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    }
    const bool shared = opt->shared_stacks != nullptr;  // 4-D: the caller owns stacks that several scans fill by slice index
    if (shared && !opt->slice_of_projection) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: shared_stacks needs slice_of_projection"};
    if (opt->write_stacks && !shared) {
      static const char* kNames[3] = {"projections_total.mha", "projections_unscattered.mha", "projections_scattered.mha"};
      for (int k = 0; k < 3; ++k) ABI_OK(mcgpu_stack_create((folder + "/" + kNames[k]).c_str(), cx, (int)nz, count, sx, sy, &stacks[k]));
    }

    std::vector<float> kms(count, 0.f);  // kernel time per projection (written before the projection is queued)
    // ---- writer thread: consumes buffers in order
    writer = std::thread([&]() {
      for (int i = 0; i < count; ++i) {
        {
          std::unique_lock<std::mutex> lk(sh.mu);
          sh.cv.wait(lk, [&] { return sh.queued > i || sh.abort; });
          if (sh.abort) return;
        }
        const int b = i & 1, p = first + i;
        if (hipSetDevice((int)dev) != hipSuccess || hipEventSynchronize(done[b]) != hipSuccess) {
          std::lock_guard<std::mutex> lk(sh.mu);
          sh.error = "!!HIP ERROR!! waiting for projection results";
          sh.abort = true;
          sh.cv.notify_all();
          return;
        }
        const double tw0 = now_s();
        int wrc = 0;
        if (shared)
          for (int k = 0; k < 3 && wrc == 0; ++k)
            wrc = mcgpu_stack_write_slice(opt->shared_stacks[k], opt->slice_of_projection[i], planes_host[b] + (size_t)k * plane);
        else if (opt->write_stacks)
          for (int k = 0; k < 3 && wrc == 0; ++k) wrc = mcgpu_stack_append(stacks[k], planes_host[b] + (size_t)k * plane);
        if (wrc == 0 && opt->write_ascii) wrc = mcgpu_write_projection(ctx, p, image_host[b], total, (double)kms[i] * 1e-3, nullptr);
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.writer_s += now_s() - tw0;
        if (wrc != 0) { sh.error = mcgpu_last_error(); sh.abort = true; }
        sh.written = i + 1;
        sh.cv.notify_all();
        if (wrc != 0) return;
      }
    });

    // ---- projection loop
    const double t0 = now_s();
    double kernel_s = 0.0, t_last_kernel = t0;
    int cur_seed = (int)seed;
    if (mode == MCGPU_MODE_COMPAT)
      for (int p = 0; p < first; ++p) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
    for (int i = 0; i < count; ++i) {
      const int b = i & 1, p = first + i;
      {  // buffer b is free once projection i-2 has been written
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return sh.written >= i - 1 || sh.abort; });
        if (sh.abort) throw ScanError{-3, sh.error};
      }
      if (nproj_all != 1 && opt->progress) {
        printf("\n\n\n   << Simulating Projection %d of %d >>\n\n\n", p + 1, (int)nproj_all);  // cbctmc/mc/simulation.py:200-219 parses this
        fflush(stdout);
      }
      ABI_OK(mcgpu_launch_projection(ctx, p, mode, cur_seed, 0, units, hpt_eff, image_dev, stream));
      if (opt->write_ascii) HIP_OK(hipMemcpyAsync(image_host[b], image_dev, words * 8, hipMemcpyDeviceToHost, stream));
      ABI_OK(mcgpu_finalize_projection(ctx, image_dev, total, cx, planes_dev[b], 1, stream));
      HIP_OK(hipMemcpyAsync(planes_host[b], planes_dev[b], 3 * plane * 4, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipEventRecord(done[b], stream));
      // kernel time of this launch; waits for the track kernel only -- its finalize and copies are already queued
      // behind it, so the next launch reaches the stream before they drain
      float ms = 0.f;
      ABI_OK(mcgpu_last_kernel_ms(ctx, &ms));
      kms[i] = ms;
      kernel_s += ms * 1e-3;
      t_last_kernel = now_s();
      {
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.queued = i + 1;
        sh.cv.notify_all();
      }
      if (mode == MCGPU_MODE_COMPAT) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
    }
    writer.join();
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      if (sh.abort) throw ScanError{-3, sh.error};
    }
    float repl[3] = {0.f, 0.f, 0.f};
    if (opt->write_stacks && !shared) {
      for (int k = 0; k < 3; ++k) {
        mcgpu_stack* s = stacks[k];
        stacks[k] = nullptr;
        ABI_OK(mcgpu_stack_finish(s, 1, &repl[k]));
      }
      if (opt->air_stack)
        ABI_OK(mcgpu_normalize_stack((folder + "/projections_total.mha").c_str(), opt->air_stack, opt->air_sigma_y, opt->air_sigma_x,
                                     (folder + "/projections_total_normalized.mha").c_str(), sx, sy));
    }
    if (report) {
      report->projections = count;
      report->histories_per_projection = total;
      report->seconds_total = now_s() - t0;
      report->seconds_kernels = kernel_s;
      report->seconds_after_last_kernel = now_s() - t_last_kernel;
      report->seconds_writer = sh.writer_s;
      for (int k = 0; k < 3; ++k) report->zero_replacement[k] = repl[k];
    }
  } catch (const ScanError& e) {
    mcgpu_set_last_error_(e.msg.c_str());
    rc = e.code ? e.code : -1;
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.abort = true;
      sh.cv.notify_all();
    }
    if (writer.joinable()) writer.join();
  }
  for (int k = 0; k < 3; ++k)
    if (stacks[k]) (void)mcgpu_stack_finish(stacks[k], 0, nullptr);  // error path: close the files
  if (stream) (void)hipStreamSynchronize(stream);
  for (int b = 0; b < 2; ++b) {
    if (planes_dev[b]) (void)hipFree(planes_dev[b]);
    if (planes_host[b]) (void)hipHostFree(planes_host[b]);
    if (image_host[b]) (void)hipHostFree(image_host[b]);
    if (done[b]) (void)hipEventDestroy(done[b]);
  }
  if (image_dev) (void)hipFree(image_dev);
  if (stream) (void)hipStreamDestroy(stream);
  return rc;
}
