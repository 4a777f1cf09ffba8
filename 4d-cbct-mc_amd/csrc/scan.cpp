// scan.cpp -- mcgpu_run_scan / mcgpu_run_scan_multi: the projection loop of main() (docker/mcgpu/MC-GPU_v1.3.cu:667-1056)
// as a device/host pipeline, written on top of the engine's own C ABI.
//
// Reference flow per projection, all on one host thread per rank: kernel -> D2H of the 45 MB tally -> MPI_Reduce to rank 0
// (:1019) -> ~0.9 s of fprintf -> re-zero.  Here every device has ONE tracking stream and everything it does is ordered on it;
// the sum of the per-device tallies is the tally exchange of exchange.cpp (the same code bench.py's ranks run, here between
// contexts of one process):
//   every device : begin(i) -> track kernel for its shard of projection i -> submit(i): devices that do not own projection i
//                  push their tally to the owner with a copy engine while projection i + 1 is tracked
//   owner of i-1 : collect(i - 1) behind its kernel i: one fused add of the landed tallies -> [ASCII text formatted on the
//                  device, only when the files are wanted] -> finalize kernel (float32 planes) -> copy of the planes (9 MB)
//                  to one of two pinned buffers -> event
//   writer thread : waits on the event, appends the planes to the three MetaImage stacks, hands the ASCII text to its worker.
// The owner rotates over the devices (projection i belongs to device i mod N; MCGPU_EXCHANGE_POLICY=0: always the first
// device, the reference's root): the only work that is exposed on a tracking stream -- the fused add, finalize -- is spread
// evenly, and so are the xGMI links.  The one-projection lag means no owner ever waits for a push (it had a whole kernel's
// time to land), and nothing ever runs BESIDE a tracking launch: a kernel that is still resident while the persistent tracking
// grid is dispatched fragments the CUs' register files for that whole launch (144 of 256 CUs then hold one workgroup instead
// of two: 9.2 -> 12.5 ms, tools/placement_probe.py).  With one device the lag is zero.
// History sharding keeps per-history RNG streams and integer tallies, so the result is identical for any number of
// devices (tests run several "devices" on one).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <strings.h>
#include <thread>
#include <vector>

#include "../../include/mcgpu_amd.h"
#include "knobs.hpp"
using mcgpu::knob_int;
using mcgpu::knob_set;
using mcgpu::knob_str;

namespace {

constexpr int kAsciiSlots = MCGPU_ASCII_SLOTS;
constexpr int kExchangeUnavailable = -7;  // the devices of a multi-device scan cannot reach each other (set-up phase only)
constexpr int kRcclUnavailable = -8;      // the RCCL route cannot be set up on these devices (set-up phase only)

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

struct DeviceLane {  // per device
  mcgpu_ctx* ctx = nullptr;
  int dev = -1;
  hipStream_t stream = nullptr;
  mcgpu_exchange* x = nullptr;                 // this device's end of the tally exchange (owns the tally buffers)
  void* planes_dev[2] = {nullptr, nullptr};    // float32 planes of the projections this device owns
  hipStream_t copy = nullptr;                  // their download: a copy engine, beside the next tracking kernel
  hipEvent_t finalized[2] = {nullptr, nullptr};  // planes_dev[b] has been written (tracking stream)
  hipEvent_t done[2] = {nullptr, nullptr};     // planes of pinned buffer b are on the host (system-scope release; copy stream)
  bool done_valid[2] = {false, false};
  unsigned long long lo = 0, hi = 0;           // shard of the units of every projection
  // RCCL route (reduce_rccl.cpp): the collective runs on a stream of its own, beside the next projection's kernel
  hipStream_t reduce = nullptr;
  void* tally[2] = {nullptr, nullptr};         // this device's tally of the projection with that parity
  hipEvent_t tracked[2] = {nullptr, nullptr};  // the tracking kernel of that projection has run (tracking stream)
  hipEvent_t reduced[2] = {nullptr, nullptr};  // the collective over that tally has run here (reduce stream)
  bool reduced_valid[2] = {false, false};
};

}  // namespace

extern "C" void mcgpu_set_last_error_(const char* message);  // engine.cpp (not part of the public ABI)

namespace {

// An input history count below 95000 is a time budget in seconds per projection (MC-GPU_v1.3.cu:650-655, :689-809: the reference
// runs a speed test and converts).  Two throw-away launches on `ctx`'s device; returns its rate in x-rays per second.
double calibrate_rate(mcgpu_ctx* ctx, int dev, int projection, int mode, long long seed, long long hpt, size_t words, void** probe_image, hipStream_t stream) {
  const unsigned long long probe = 4000000ULL;
  const unsigned long long probe_units = mode == MCGPU_MODE_COMPAT ? (probe + (unsigned long long)hpt - 1) / (unsigned long long)hpt : probe;
  float ms = 0.f;
  HIP_OK(hipSetDevice(dev));
  if (!*probe_image) HIP_OK(hipMalloc(probe_image, words * 8));
  HIP_OK(hipMemsetAsync(*probe_image, 0, words * 8, stream));
  for (int rep = 0; rep < 2; ++rep) {  // the first launch pays one-off costs
    ABI_OK(mcgpu_launch_projection(ctx, projection, mode, (int)seed, 0, probe_units, (int)hpt, *probe_image, stream));
    ABI_OK(mcgpu_last_kernel_ms(ctx, &ms));
  }
  ABI_OK(mcgpu_dose_clear(ctx));
  return (double)probe / (ms > 0.f ? ms * 1e-3 : 1e-3);
}

// One scan over n_ctx devices that share every projection's histories (n_ctx = 1: the plain single-device pipeline).
// `use_rccl`: the per-device tallies of a projection are summed by one ncclReduce to the projection's owner (reduce_rccl.cpp) instead
// of through the tally exchange.
int run_scan_sharing_histories(mcgpu_ctx* const* ctxs, int n_ctx, const mcgpu_scan_options* opt, mcgpu_scan_report* report, bool use_rccl = false) {
  std::vector<DeviceLane> D((size_t)n_ctx);
  mcgpu_rccl* rccl = nullptr;
  float* planes_host[2] = {nullptr, nullptr};
  uint64_t* image_host[2] = {nullptr, nullptr};
  void* probe_image = nullptr;  // tally of the throw-away launches (time calibration, preset choice)
  std::vector<unsigned char> mailboxes;  // host region of the exchange (contexts of one process: plain memory)
  mcgpu_stack* stacks[3] = {nullptr, nullptr, nullptr};
  std::thread writer;
  struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    int queued = 0, written = 0;  // projections handed to / finished by the writer
    bool abort = false;
    std::string error;
    double writer_s = 0.0;
    // ASCII files of the device formatter: one worker per slot, so that several files are written side by side (writes
    // to ONE file do not scale with threads -- they serialise on its inode lock -- writes to different files do)
    bool ascii_busy[kAsciiSlots] = {};
    int ascii_p[kAsciiSlots] = {};
    mcgpu_ctx* ascii_ctx[kAsciiSlots] = {};  // the context (device) that formatted the slot's text
    double ascii_seconds[kAsciiSlots] = {};
    bool ascii_quit = false;
  } sh;
  std::thread ascii_worker[kAsciiSlots];
  int n_ascii = kAsciiSlots;  // formatter slots in use = files written side by side
  mcgpu_ctx* ctx = ctxs[0];
  int rc = 0;
  try {
    long long nproj_all = 1, hist_in = 0, seed = 0, tpb = 128, hpt = 150, nx = 0, nz = 0;
    ABI_OK(mcgpu_config_i64(ctx, "num_projections", &nproj_all));
    ABI_OK(mcgpu_config_i64(ctx, "total_histories", &hist_in));
    ABI_OK(mcgpu_config_i64(ctx, "seed", &seed));
    ABI_OK(mcgpu_config_i64(ctx, "threads_per_block", &tpb));
    ABI_OK(mcgpu_config_i64(ctx, "histories_per_thread", &hpt));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_x", &nx));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_z", &nz));
    for (int g = 0; g < n_ctx; ++g) {
      long long dev = -1;
      if (!ctxs[g]) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: null context"};
      ABI_OK(mcgpu_config_i64(ctxs[g], "device_id", &dev));
      if (dev < 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: a context has no device"};
      D[g].ctx = ctxs[g];
      D[g].dev = (int)dev;
    }
    double px_x = 0, px_z = 0;
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_x_mm", &px_x));
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_z_mm", &px_z));
    const int mode = opt->mode == MCGPU_MODE_COMPAT ? MCGPU_MODE_COMPAT : (opt->mode == MCGPU_MODE_FAST_F64 ? MCGPU_MODE_FAST_F64 : MCGPU_MODE_FAST);
    const int first = opt->first_projection > 0 ? opt->first_projection : 0;
    const int range = (opt->num_projections > 0) ? opt->num_projections : (int)nproj_all - first;
    if (first + range > nproj_all || range <= 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: projection range outside the trajectory"};
    // projections outside the input's angular region of interest are not simulated (MC-GPU_v1.3.cu:670-677; like the
    // reference, the test uses initial_angle + p * D_angle even when specific angles are given)
    double d_angle = -1.0, angle0 = 0.0, roi0 = 0.0, roi1 = 0.0;
    ABI_OK(mcgpu_config_f64(ctx, "D_angle", &d_angle));
    ABI_OK(mcgpu_config_f64(ctx, "initial_angle", &angle0));
    ABI_OK(mcgpu_config_f64(ctx, "angularROI_0", &roi0));
    ABI_OK(mcgpu_config_f64(ctx, "angularROI_1", &roi1));
    auto outside_roi = [&](int p) { const double a = angle0 + p * d_angle; return a < roi0 || a > roi1; };
    // offsets within the range of the projections this scan simulates, and their ordinal among ALL simulated projections of the
    // range (a projection-sharded scan takes every stride-th of them: the COMPAT seed still moves on once per simulated
    // projection, whoever simulates it)
    const int stride = opt->projection_stride > 1 ? opt->projection_stride : 1, phase = stride > 1 ? opt->projection_phase : 0;
    if (phase < 0 || phase >= stride) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: projection_phase outside [0, projection_stride)"};
    std::vector<int> sim, ord;
    {
      int n_sim = 0;
      for (int k = 0; k < range; ++k)
        if (!outside_roi(first + k)) {
          if (n_sim % stride == phase) { sim.push_back(k); ord.push_back(n_sim); }
          ++n_sim;
        }
    }
    const int count = (int)sim.size();
    unsigned long long H = opt->histories_per_projection ? opt->histories_per_projection : (unsigned long long)hist_in;
    // An input value below 95000 is a time budget in seconds per projection, not a history count (MC-GPU_v1.3.cu:650-655,
    // :689-809: the reference runs a speed test and converts).  Calibrate on device 0 with a throw-away launch.
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

    // ---- device resources
    // pinned buffers are read by the writer thread and filled from whichever device owns the projection: portable
    const unsigned int pinned_flags = (knob_set("MCGPU_PINNED_COHERENT") ? hipHostMallocDefault : hipHostMallocNonCoherent) | hipHostMallocPortable;
    const bool ascii_on_host = knob_set("MCGPU_ASCII_HOST");  // A/B: the threaded host formatter of report.cpp
    if (const char* w = knob_str("MCGPU_ASCII_WRITERS")) n_ascii = std::min(std::max(atoi(w), 1), kAsciiSlots);
    const bool single = (n_ctx == 1);
    int policy = MCGPU_EXCHANGE_LOCAL | MCGPU_EXCHANGE_ROTATE;
    if (const char* v = knob_str("MCGPU_EXCHANGE_POLICY")) policy = MCGPU_EXCHANGE_LOCAL | (atoi(v) ? MCGPU_EXCHANGE_ROTATE : 0);
    mailboxes.assign(mcgpu_exchange_shared_bytes(n_ctx) * (size_t)(use_rccl ? n_ctx : 1), 0);
    void* const mailbox_region = mailboxes.data();
    const bool rotate_owner = (policy & MCGPU_EXCHANGE_ROTATE) != 0;
    // owner of projection j of this scan: the device that holds the summed tally, finalizes and downloads it
    auto owner_of = [&](int j) { return use_rccl ? (rotate_owner ? j % n_ctx : 0) : mcgpu_exchange_owner(D[0].x, j); };
    for (int g = 0; g < n_ctx; ++g) {
      HIP_OK(hipSetDevice(D[g].dev));
      HIP_OK(hipStreamCreate(&D[g].stream));
      if (use_rccl) {
        // every device keeps its double-buffered tally in an exchange end of its own (a world of one: begin / submit / collect stay local)
        ABI_OK(mcgpu_exchange_create(D[g].dev, 0, 1, words, MCGPU_EXCHANGE_LOCAL, mailboxes.data() + (size_t)g * mcgpu_exchange_shared_bytes(n_ctx), &D[g].x));
        HIP_OK(hipStreamCreateWithFlags(&D[g].reduce, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
          HIP_OK(hipEventCreateWithFlags(&D[g].tracked[b], hipEventDisableTiming));
          HIP_OK(hipEventCreateWithFlags(&D[g].reduced[b], hipEventDisableTiming));
        }
      } else {
        ABI_OK(mcgpu_exchange_create(D[g].dev, g, n_ctx, words, policy, mailbox_region, &D[g].x));
      }
    }
    if (use_rccl) {
      std::vector<int> devs((size_t)n_ctx);
      for (int g = 0; g < n_ctx; ++g) devs[(size_t)g] = D[g].dev;
      if (mcgpu_rccl_create(devs.data(), n_ctx, &rccl) != 0) throw ScanError{kRcclUnavailable, mcgpu_last_error()};
    } else try {
      // peer access between every pair of devices, and one small copy-engine transfer into every peer's landing buffer: a node
      // on which either fails is reported with kExchangeUnavailable BEFORE anything has been simulated or written, and
      // mcgpu_run_scan_multi then shards the scan by projection instead (same output bytes, nothing crosses between devices)
      if (n_ctx > 1 && knob_set("MCGPU_EXCHANGE_FAIL_PROBE")) throw ScanError{-1, "!!ERROR!! tally exchange: probe failure requested (MCGPU_EXCHANGE_FAIL_PROBE)"};  // test hook
      for (int g = 0; g < n_ctx; ++g)
        for (int h = 0; h < n_ctx; ++h)
          if (g != h) ABI_OK(mcgpu_exchange_connect_local(D[g].x, D[h].x));
      for (int g = 0; g < n_ctx; ++g) ABI_OK(mcgpu_exchange_probe(D[g].x));
    } catch (const ScanError& e) {
      throw ScanError{kExchangeUnavailable, e.msg};
    }
    HIP_OK(hipSetDevice(D[0].dev));
    if (by_time) {
      const double rate = calibrate_rate(ctx, D[0].dev, first, mode, seed, hpt, words, &probe_image, D[0].stream) * n_ctx;
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
    // FAST batching thresholds: a few presets timed with throw-away launches on the first simulated projection (the best
    // one differs between geometries by 5-7 %; the tallies do not depend on the choice).  Skipped for short scans and when
    // the environment pins the knobs.
    long long fast_scheduler = 0;  // the presets are per-wave pool thresholds; the workgroup-level pool keeps its own defaults
    ABI_OK(mcgpu_config_i64(ctx, "fast_scheduler", &fast_scheduler));
    if (mode != MCGPU_MODE_COMPAT && fast_scheduler == 0 && total >= 20000000ULL && !knob_set("MCGPU_THRESH_COMPTON") && !knob_set("MCGPU_THRESH_NEW") &&
        !knob_set("MCGPU_SWAP_BATCH") && !knob_set("MCGPU_NO_AUTOTUNE")) {
      static const int presets[3][5] = {{40, 12, 44, 12, 40}, {32, 8, 36, 12, 40}, {36, 16, 44, 12, 44}};  // profiles/r05o_*, r05p_*
      const unsigned long long probe = 6000000ULL;
      if (!probe_image) HIP_OK(hipMalloc(&probe_image, words * 8));
      HIP_OK(hipMemsetAsync(probe_image, 0, words * 8, D[0].stream));
      int best = 0;
      float best_ms = 1e30f;
      for (int c = 0; c < 3; ++c) {
        ABI_OK(mcgpu_set_fast_schedule(ctx, presets[c][0], presets[c][1], presets[c][2], presets[c][3], presets[c][4]));
        float ms = 0.f, fastest = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {  // the first launch of a preset pays the parameter upload
          ABI_OK(mcgpu_launch_projection(ctx, first + (count > 0 ? sim[0] : 0), mode, (int)seed, 0, probe, (int)hpt, probe_image, D[0].stream));
          ABI_OK(mcgpu_last_kernel_ms(ctx, &ms));
          if (rep > 0 && ms < fastest) fastest = ms;
        }
        if (fastest < best_ms) { best_ms = fastest; best = c; }
      }
      for (int g = 0; g < n_ctx; ++g)
        ABI_OK(mcgpu_set_fast_schedule(D[g].ctx, presets[best][0], presets[best][1], presets[best][2], presets[best][3], presets[best][4]));
      HIP_OK(hipSetDevice(D[0].dev));
      ABI_OK(mcgpu_dose_clear(ctx));
      if (opt->progress) {
        printf("       FAST batching preset %d of 3 (thresholds %d/%d/%d, %d, %d)\n", best + 1, presets[best][0], presets[best][1], presets[best][2],
               presets[best][3], presets[best][4]);
        fflush(stdout);
      }
    }
    for (int g = 0; g < n_ctx; ++g) { D[g].lo = units * g / n_ctx; D[g].hi = units * (g + 1) / n_ctx; }
    if (probe_image) {
      HIP_OK(hipStreamSynchronize(D[0].stream));
      HIP_OK(hipFree(probe_image));
      probe_image = nullptr;
    }
    for (int g = 0; g < n_ctx; ++g) {  // every device that can own a projection finalizes it
      if (g > 0 && !(policy & MCGPU_EXCHANGE_ROTATE)) break;
      HIP_OK(hipSetDevice(D[g].dev));
      HIP_OK(hipStreamCreateWithFlags(&D[g].copy, hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) {
        HIP_OK(hipEventCreateWithFlags(&D[g].finalized[b], hipEventDisableTiming));
        HIP_OK(hipMalloc(&D[g].planes_dev[b], 3 * plane * 4));
        // the writer thread reads non-coherent pinned memory after waiting on this event: that needs a SYSTEM-scope release,
        // which a default event does not promise (device scope only)
        HIP_OK(hipEventCreateWithFlags(&D[g].done[b], hipEventDisableTiming | hipEventReleaseToSystem));
      }
    }
    HIP_OK(hipSetDevice(D[0].dev));
    for (int b = 0; b < 2; ++b) {
      // non-coherent (CPU-cacheable) pinned memory: the writer thread reads every byte (ordering: see the event above)
      HIP_OK(hipHostMalloc((void**)&planes_host[b], 3 * plane * 4, pinned_flags));
      if (opt->write_ascii && ascii_on_host) HIP_OK(hipHostMalloc((void**)&image_host[b], words * 8, pinned_flags));
    }
    const bool shared = opt->shared_stacks != nullptr;  // 4-D: the caller owns stacks that several scans fill by slice index
    if (shared && !opt->slice_of_projection) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: shared_stacks needs slice_of_projection"};
    if (opt->write_stacks && !shared && count > 0) {
      static const char* kNames[3] = {"projections_total.mha", "projections_unscattered.mha", "projections_scattered.mha"};
      for (int k = 0; k < 3; ++k) ABI_OK(mcgpu_stack_create((folder + "/" + kNames[k]).c_str(), cx, (int)nz, count, sx, sy, &stacks[k]));
    }

    std::vector<float> kms(count, 0.f);  // kernel time per projection (written before the projection is queued)
    const bool ascii_async = opt->write_ascii && !ascii_on_host;
    if (ascii_async)
      for (int b = 0; b < n_ascii; ++b)
        ascii_worker[b] = std::thread([&, b]() {
          for (;;) {
            int p;
            double secs;
            mcgpu_ctx* fctx;
            {
              std::unique_lock<std::mutex> lk(sh.mu);
              sh.cv.wait(lk, [&] { return sh.ascii_busy[b] || sh.ascii_quit || sh.abort; });
              if (!sh.ascii_busy[b]) return;
              p = sh.ascii_p[b];
              secs = sh.ascii_seconds[b];
              fctx = sh.ascii_ctx[b];
            }
            const double tw0 = now_s();
            const int wrc = mcgpu_write_formatted_projection(fctx, p, b, total, secs, nullptr);
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.writer_s += now_s() - tw0;
            if (wrc != 0) { sh.error = mcgpu_last_error(); sh.abort = true; }
            sh.ascii_busy[b] = false;
            sh.cv.notify_all();
            if (wrc != 0) return;
          }
        });
    // ---- writer thread: consumes buffers in order
    writer = std::thread([&]() {
      for (int i = 0; i < count; ++i) {
        {
          std::unique_lock<std::mutex> lk(sh.mu);
          sh.cv.wait(lk, [&] { return sh.queued > i || sh.abort; });
          if (sh.abort) return;
        }
        const int b = i & 1, p = first + sim[i], o = owner_of(i);
        if (hipSetDevice(D[o].dev) != hipSuccess || hipEventSynchronize(D[o].done[b]) != hipSuccess) {
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
            wrc = mcgpu_stack_write_slice(opt->shared_stacks[k], opt->slice_of_projection[sim[i]], planes_host[b] + (size_t)k * plane);
        else if (opt->write_stacks) {
          // the three stacks side by side (each append scans its plane for zeros and copies it into the page cache)
          int rc3[3] = {0, 0, 0};
          std::string err3[3];
          std::thread side[2];
          for (int k = 1; k < 3; ++k)
            side[k - 1] = std::thread([&, k] {
              rc3[k] = mcgpu_stack_append(stacks[k], planes_host[b] + (size_t)k * plane);
              if (rc3[k] != 0) err3[k] = mcgpu_last_error();  // the last error is per thread
            });
          rc3[0] = mcgpu_stack_append(stacks[0], planes_host[b]);
          for (auto& t : side) t.join();
          for (int k = 2; k >= 0; --k)
            if (rc3[k] != 0) { wrc = rc3[k]; if (k > 0) mcgpu_set_last_error_(err3[k].c_str()); }
        }
        if (wrc == 0 && opt->write_ascii && ascii_on_host) wrc = mcgpu_write_projection(ctx, p, image_host[b], total, (double)kms[i] * 1e-3, nullptr);
        std::lock_guard<std::mutex> lk(sh.mu);
        if (wrc == 0 && ascii_async) {  // the slot is free: the projection loop waited for that before it formatted into it
          const int a = i % n_ascii;
          sh.ascii_p[a] = p;
          sh.ascii_ctx[a] = D[o].ctx;
          sh.ascii_seconds[a] = (double)kms[i] * 1e-3;
          sh.ascii_busy[a] = true;
        }
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
    if (mode == MCGPU_MODE_COMPAT) {  // the seed moves on per SIMULATED projection (MC-GPU_v1.3.cu:869)
      for (int p = 0; p < first; ++p)
        if (!outside_roi(p)) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
      for (int k = 0; k < (count > 0 ? ord[0] : 0); ++k) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
    }
    // sum + finalize of projection j on its owner's stream, then hand it to the writer
    auto enqueue_reduce = [&](int j) {
      const int b = j & 1, o = owner_of(j);
      {  // pinned buffer b is free once projection j-2 has been written.  Formatter slot j % n_ascii was last used by
         // projection j - n_ascii: that one must have been HANDED to its worker (the writer sets ascii_busy when it has
         // written the projection's stacks, together with `written`) and the worker must be done.  With one slot the
         // previous projection itself is the slot's last user, so `written >= j - 1` is not enough (it would let
         // projection j be formatted over the text of j - 1 before j - 1 was even handed over).
        const int handed = j - std::min(1, n_ascii - 1);
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return (sh.written >= handed && !sh.ascii_busy[j % n_ascii]) || sh.abort; });
        if (sh.abort) throw ScanError{-3, sh.error};
      }
      if (use_rccl) {
        // one ncclReduce(uint64, sum, root = owner) over the devices' tallies of projection j (the reference's MPI_Reduce,
        // MC-GPU_v1.3.cu:1019), on the reduce streams: behind kernel j of each device, beside its kernel j + 1
        std::vector<void*> bufs((size_t)n_ctx), streams((size_t)n_ctx);
        for (int g = 0; g < n_ctx; ++g) {
          HIP_OK(hipSetDevice(D[g].dev));
          HIP_OK(hipStreamWaitEvent(D[g].reduce, D[g].tracked[b], 0));
          bufs[(size_t)g] = D[g].tally[b];
          streams[(size_t)g] = (void*)D[g].reduce;
        }
        ABI_OK(mcgpu_rccl_reduce_u64(rccl, bufs.data(), words, o, streams.data()));
        for (int g = 0; g < n_ctx; ++g) {
          HIP_OK(hipSetDevice(D[g].dev));
          HIP_OK(hipEventRecord(D[g].reduced[b], D[g].reduce));
          D[g].reduced_valid[b] = true;
        }
        HIP_OK(hipSetDevice(D[o].dev));
        HIP_OK(hipStreamWaitEvent(D[o].stream, D[o].reduced[b], 0));  // the owner's tracking stream goes on with the summed tally
        for (int g = 0; g < n_ctx; ++g)
          if (g != o) { void* unused = nullptr; ABI_OK(mcgpu_exchange_collect(D[g].x, j, D[g].stream, &unused)); }  // bookkeeping of the local ends
      }
      HIP_OK(hipSetDevice(D[o].dev));
      hipStream_t const so = D[o].stream;
      void* tally = nullptr;
      ABI_OK(mcgpu_exchange_collect(D[o].x, j, so, &tally));  // exchange: the landed tallies of the other devices, added in one pass
      if (!tally) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: the owner of a projection got no tally"};
      // the reference's ASCII file: its 63 MB of text are formatted on the device (ascii_device.hip) while the tally is
      // there; the writer thread downloads and writes them while the next projection is tracked
      if (opt->write_ascii) {
        if (ascii_on_host) HIP_OK(hipMemcpyAsync(image_host[b], tally, words * 8, hipMemcpyDeviceToHost, so));
        else ABI_OK(mcgpu_format_projection(D[o].ctx, tally, total, j % n_ascii, so));
      }
      // planes_dev[b] was last downloaded two projections ago (by this device, if it owned that one)
      if (D[o].done_valid[b]) HIP_OK(hipStreamWaitEvent(so, D[o].done[b], 0));
      ABI_OK(mcgpu_finalize_projection(D[o].ctx, tally, total, cx, D[o].planes_dev[b], 0, so));  // begin() zeroes the buffer for its next user
      // the 9 MB of planes go to the host on a copy engine, beside the next tracking kernel (on the tracking stream the copy
      // held the next launch back by 0.17 ms per projection)
      HIP_OK(hipEventRecord(D[o].finalized[b], so));
      HIP_OK(hipStreamWaitEvent(D[o].copy, D[o].finalized[b], 0));
      HIP_OK(hipMemcpyAsync(planes_host[b], D[o].planes_dev[b], 3 * plane * 4, hipMemcpyDeviceToHost, D[o].copy));
      HIP_OK(hipEventRecord(D[o].done[b], D[o].copy));
      D[o].done_valid[b] = true;
    };
    auto publish = [&](int j) {  // kms[j] is final: the writer may take projection j
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.queued = j + 1;
      sh.cv.notify_all();
    };
    const double kRad2Deg = 180.0 / 3.14159265358979323846;
    auto print_skipped = [&](int from, int to) {  // offsets [from, to) of the range
      for (int k = from; k < to && opt->progress && phase == 0; ++k)
        if (outside_roi(first + k))
          printf("         << Skipping projection #%d of %d >> Angle %f degrees: outside angular region of interest.\n", first + k + 1, (int)nproj_all,
               (angle0 + (first + k) * d_angle) * kRad2Deg);
    };
    for (int i = 0; i < count; ++i) {
      const int p = first + sim[i];
      print_skipped(i ? sim[i - 1] + 1 : 0, sim[i]);
      if (nproj_all != 1 && opt->progress) {
        printf("\n\n\n   << Simulating Projection %d of %d >>\n\n\n", p + 1, (int)nproj_all);  // cbctmc/mc/simulation.py:200-219 parses this
        fflush(stdout);
      }
      // every device tracks its shard into a zeroed tally buffer; the devices that do not own the projection then push theirs
      // to the owner (copy engine, beside the next projection's kernel)
      for (int g = 0; g < n_ctx; ++g) {
        HIP_OK(hipSetDevice(D[g].dev));
        void* tally = nullptr;
        if (use_rccl && D[g].reduced_valid[i & 1]) HIP_OK(hipStreamWaitEvent(D[g].stream, D[g].reduced[i & 1], 0));  // begin() zeroes the buffer the collective of i - 2 read
        ABI_OK(mcgpu_exchange_begin(D[g].x, i, D[g].stream, &tally));
        ABI_OK(mcgpu_launch_projection(D[g].ctx, p, mode, cur_seed, D[g].lo, D[g].hi - D[g].lo, hpt_eff, tally, D[g].stream));
        ABI_OK(mcgpu_exchange_submit(D[g].x, i, D[g].stream));
        if (use_rccl) {
          D[g].tally[i & 1] = tally;
          HIP_OK(hipEventRecord(D[g].tracked[i & 1], D[g].stream));
        }
      }
      // the owner of the previous projection, behind its own kernel i (one device: this projection)
      if (single) enqueue_reduce(i);
      else if (i >= 1) { enqueue_reduce(i - 1); publish(i - 1); }
      // kernel time of this projection = the slowest device's launch; waiting for it also paces the host
      float ms = 0.f;
      for (int g = 0; g < n_ctx; ++g) {
        float m = 0.f;
        ABI_OK(mcgpu_last_kernel_ms(D[g].ctx, &m));
        ms = m > ms ? m : ms;
      }
      kms[i] = ms;
      kernel_s += ms * 1e-3;
      t_last_kernel = now_s();
      if (single) publish(i);
      if (mode == MCGPU_MODE_COMPAT)
        for (int k = ord[i]; k < (i + 1 < count ? ord[i + 1] : ord[i] + 1); ++k) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
    }
    print_skipped(count ? sim[count - 1] + 1 : 0, range);
    if (!single && count > 0) { enqueue_reduce(count - 1); publish(count - 1); }
    writer.join();
    {
      std::unique_lock<std::mutex> lk(sh.mu);
      sh.cv.wait(lk, [&] {
        bool busy = false;
        for (bool b : sh.ascii_busy) busy = busy || b;
        return !busy || sh.abort;
      });
      sh.ascii_quit = true;
      sh.cv.notify_all();
    }
    for (auto& w : ascii_worker)
      if (w.joinable()) w.join();
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      if (sh.abort) throw ScanError{-3, sh.error};
    }
    float repl[3] = {0.f, 0.f, 0.f};
    if (opt->write_stacks && !shared && count > 0) {
      // zero replacement and close of the three stacks side by side (page-cache work on three different files)
      int rc3[3] = {0, 0, 0};
      std::string err3[3];
      std::thread fin[3];
      for (int k = 0; k < 3; ++k) {
        mcgpu_stack* s = stacks[k];
        stacks[k] = nullptr;
        fin[k] = std::thread([&, k, s] {
          rc3[k] = mcgpu_stack_finish(s, 1, &repl[k]);
          if (rc3[k] != 0) err3[k] = mcgpu_last_error();  // the last error is per thread
        });
      }
      for (auto& t : fin) t.join();
      for (int k = 0; k < 3; ++k)
        if (rc3[k] != 0) throw ScanError{rc3[k], err3[k]};
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
      report->kernel_ms_min = report->kernel_ms_max = count > 0 ? (double)kms[0] : 0.0;
      for (int i = 1; i < count; ++i) {
        report->kernel_ms_min = std::min(report->kernel_ms_min, (double)kms[i]);
        report->kernel_ms_max = std::max(report->kernel_ms_max, (double)kms[i]);
      }
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
    for (auto& w : ascii_worker)
      if (w.joinable()) w.join();
  }
  for (int k = 0; k < 3; ++k)
    if (stacks[k]) (void)mcgpu_stack_finish(stacks[k], 0, nullptr);  // error path: close the files
  for (auto& d : D) {
    if (d.dev < 0) continue;
    (void)hipSetDevice(d.dev);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.copy) (void)hipStreamSynchronize(d.copy);
    if (d.reduce) (void)hipStreamSynchronize(d.reduce);
  }
  if (rccl) mcgpu_rccl_destroy(rccl);
  if (D[0].dev >= 0) {
    (void)hipSetDevice(D[0].dev);
    if (probe_image) (void)hipFree(probe_image);
    for (int b = 0; b < 2; ++b) {
      if (planes_host[b]) (void)hipHostFree(planes_host[b]);
      if (image_host[b]) (void)hipHostFree(image_host[b]);
    }
  }
  for (auto& d : D) {
    if (d.dev < 0) continue;
    (void)hipSetDevice(d.dev);
    for (int b = 0; b < 2; ++b) {
      if (d.planes_dev[b]) (void)hipFree(d.planes_dev[b]);
      if (d.done[b]) (void)hipEventDestroy(d.done[b]);
      if (d.finalized[b]) (void)hipEventDestroy(d.finalized[b]);
    }
    if (d.copy) (void)hipStreamDestroy(d.copy);
    if (d.reduce) (void)hipStreamDestroy(d.reduce);
    for (int b = 0; b < 2; ++b) {
      if (d.tracked[b]) (void)hipEventDestroy(d.tracked[b]);
      if (d.reduced[b]) (void)hipEventDestroy(d.reduced[b]);
    }
  }
  for (auto& d : D)  // after every device has drained: an exchange end frees memory its peers push into
    if (d.x) mcgpu_exchange_destroy(d.x);
  for (auto& d : D) {
    if (d.dev < 0) continue;
    (void)hipSetDevice(d.dev);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  return rc;
}

// Projection sharding (SURVEY.md 8e's fallback mode; the reference has no counterpart -- its ranks always share a projection,
// MC-GPU_v1.3.cu:913-1019): context g runs the single-device pipeline over the simulated projections number g, g + n, ... on a
// thread of its own, with all of their histories.  Nothing crosses between the devices: no exchange, no peer access, no
// collective.  The MetaImage stacks are shared and filled by slice index; ASCII files are per projection anyway.  Every output
// byte equals a one-device scan's (per-history streams / per-projection seeds do not depend on who simulates a projection).
int run_scan_sharing_projections(mcgpu_ctx* const* ctxs, int n_ctx, const mcgpu_scan_options* opt, mcgpu_scan_report* report) {
  mcgpu_stack* stacks[3] = {nullptr, nullptr, nullptr};
  int rc = 0;
  std::string error;
  try {
    mcgpu_ctx* ctx = ctxs[0];
    long long nproj_all = 1, nx = 0, nz = 0;
    ABI_OK(mcgpu_config_i64(ctx, "num_projections", &nproj_all));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_x", &nx));
    ABI_OK(mcgpu_config_i64(ctx, "num_pixels_z", &nz));
    double px_x = 0, px_z = 0, d_angle = -1.0, angle0 = 0.0, roi0 = 0.0, roi1 = 0.0;
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_x_mm", &px_x));
    ABI_OK(mcgpu_config_f64(ctx, "pixel_size_z_mm", &px_z));
    ABI_OK(mcgpu_config_f64(ctx, "D_angle", &d_angle));
    ABI_OK(mcgpu_config_f64(ctx, "initial_angle", &angle0));
    ABI_OK(mcgpu_config_f64(ctx, "angularROI_0", &roi0));
    ABI_OK(mcgpu_config_f64(ctx, "angularROI_1", &roi1));
    const int first = opt->first_projection > 0 ? opt->first_projection : 0;
    const int range = (opt->num_projections > 0) ? opt->num_projections : (int)nproj_all - first;
    if (first + range > nproj_all || range <= 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: projection range outside the trajectory"};
    if (opt->projection_stride > 1) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan_multi: MCGPU_SHARD_PROJECTIONS sets the projection stride itself"};
    // slice of every offset of the range: its ordinal among the simulated projections (the order a one-device scan appends in)
    std::vector<int> slice_of((size_t)range, -1);
    int count = 0;
    for (int k = 0; k < range; ++k) {
      const double a = angle0 + (first + k) * d_angle;
      if (!(a < roi0 || a > roi1)) slice_of[(size_t)k] = count++;
    }
    const int cx = (opt->crop_nx > 0 && opt->crop_nx < nx) ? opt->crop_nx : (int)nx;
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
    const bool own_stacks = opt->write_stacks && !opt->shared_stacks && count > 0;
    if (own_stacks) {
      static const char* kNames[3] = {"projections_total.mha", "projections_unscattered.mha", "projections_scattered.mha"};
      for (int k = 0; k < 3; ++k) ABI_OK(mcgpu_stack_create((folder + "/" + kNames[k]).c_str(), cx, (int)nz, count, sx, sy, &stacks[k]));
    }
    std::vector<mcgpu_scan_options> o((size_t)n_ctx, *opt);
    // a time-limited run (history count below 95000 = seconds per projection): ONE calibration, on device 0, for all the shards
    // -- each device calibrating by itself would simulate its own history count and the files would depend on who wrote them
    {
      long long hist_in = 0, seed = 0, hpt = 150, dev0 = -1;
      ABI_OK(mcgpu_config_i64(ctx, "total_histories", &hist_in));
      ABI_OK(mcgpu_config_i64(ctx, "seed", &seed));
      ABI_OK(mcgpu_config_i64(ctx, "histories_per_thread", &hpt));
      ABI_OK(mcgpu_config_i64(ctx, "device_id", &dev0));
      if (!opt->histories_per_projection && hist_in < 95000 && count > 0) {
        if (dev0 < 0) throw ScanError{-1, "!!ERROR!! mcgpu_run_scan: a context has no device"};
        const int mode = opt->mode == MCGPU_MODE_COMPAT ? MCGPU_MODE_COMPAT : (opt->mode == MCGPU_MODE_FAST_F64 ? MCGPU_MODE_FAST_F64 : MCGPU_MODE_FAST);
        void* probe_image = nullptr;
        double rate = 0.0;
        try {
          rate = calibrate_rate(ctx, (int)dev0, first, mode, seed, hpt, (size_t)4 * nx * nz, &probe_image, nullptr);
          HIP_OK(hipDeviceSynchronize());
        } catch (...) {
          if (probe_image) (void)hipFree(probe_image);
          throw;
        }
        (void)hipFree(probe_image);
        unsigned long long H = (unsigned long long)(rate * (double)hist_in);  // a device simulates a projection alone: its own rate
        if (H < 100000ULL) H = 100000ULL;
        for (int g = 0; g < n_ctx; ++g) o[(size_t)g].histories_per_projection = H;
        if (opt->progress) {
          printf("       Time-limited run: %lld s per projection at %.3e x-rays/s per device -> %llu histories per projection\n", hist_in, rate, H);
          fflush(stdout);
        }
      }
    }
    std::vector<mcgpu_scan_report> r((size_t)n_ctx);
    std::vector<int> rcs((size_t)n_ctx, 0);
    std::vector<std::string> errs((size_t)n_ctx);
    std::vector<std::thread> th;
    const double t0 = now_s();
    for (int g = 0; g < n_ctx; ++g) {
      memset(&r[(size_t)g], 0, sizeof(mcgpu_scan_report));
      o[(size_t)g].shard = MCGPU_SHARD_HISTORIES;  // a single-context scan
      o[(size_t)g].projection_stride = n_ctx;
      o[(size_t)g].projection_phase = g;
      o[(size_t)g].air_stack = nullptr;  // normalisation runs once, below, over the finished stack
      if (own_stacks) {
        o[(size_t)g].shared_stacks = stacks;
        o[(size_t)g].slice_of_projection = slice_of.data();
      }
      th.emplace_back([&, g] {
        rcs[(size_t)g] = run_scan_sharing_histories(&ctxs[g], 1, &o[(size_t)g], &r[(size_t)g]);
        if (rcs[(size_t)g] != 0) errs[(size_t)g] = mcgpu_last_error();  // the last error is per thread
      });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < n_ctx; ++g)
      if (rcs[(size_t)g] != 0) throw ScanError{rcs[(size_t)g], errs[(size_t)g]};
    float repl[3] = {0.f, 0.f, 0.f};
    if (own_stacks) {
      for (int k = 0; k < 3; ++k) {
        mcgpu_stack* st = stacks[k];
        stacks[k] = nullptr;
        ABI_OK(mcgpu_stack_finish(st, 1, &repl[k]));
      }
      if (opt->air_stack)
        ABI_OK(mcgpu_normalize_stack((folder + "/projections_total.mha").c_str(), opt->air_stack, opt->air_sigma_y, opt->air_sigma_x,
                                     (folder + "/projections_total_normalized.mha").c_str(), sx, sy));
    }
    if (report) {
      memset(report, 0, sizeof *report);
      report->histories_per_projection = r[0].histories_per_projection;
      report->seconds_total = now_s() - t0;
      report->kernel_ms_min = 1e30;
      for (int g = 0; g < n_ctx; ++g) {
        if (r[(size_t)g].projections > 0) {
          report->kernel_ms_min = std::min(report->kernel_ms_min, r[(size_t)g].kernel_ms_min);
          report->kernel_ms_max = std::max(report->kernel_ms_max, r[(size_t)g].kernel_ms_max);
        }
        report->projections += r[(size_t)g].projections;
        report->seconds_kernels = std::max(report->seconds_kernels, r[(size_t)g].seconds_kernels);  // the devices run side by side
        report->seconds_after_last_kernel = std::max(report->seconds_after_last_kernel, r[(size_t)g].seconds_after_last_kernel);
        report->seconds_writer += r[(size_t)g].seconds_writer;
      }
      if (report->projections == 0) report->kernel_ms_min = 0.0;
      for (int k = 0; k < 3; ++k) report->zero_replacement[k] = repl[k];
    }
  } catch (const ScanError& e) {
    error = e.msg;
    rc = e.code ? e.code : -1;
  }
  for (int k = 0; k < 3; ++k)
    if (stacks[k]) (void)mcgpu_stack_finish(stacks[k], 0, nullptr);  // error path: close the files
  if (rc != 0) mcgpu_set_last_error_(error.c_str());
  return rc;
}

}  // namespace

extern "C" int mcgpu_run_scan_multi(mcgpu_ctx* const* ctxs, int n_ctx, const mcgpu_scan_options* caller_opt, mcgpu_scan_report* report) {
  if (!ctxs || n_ctx < 1 || !caller_opt || !ctxs[0]) { mcgpu_set_last_error_("!!ERROR!! mcgpu_run_scan: null argument"); return -1; }
  // a caller built against an older header passes a shorter struct: what it does not have reads as zero
  if (caller_opt->struct_size < sizeof(unsigned int) + sizeof(int)) {
    mcgpu_set_last_error_("!!ERROR!! mcgpu_run_scan: set mcgpu_scan_options.struct_size = sizeof(mcgpu_scan_options)");
    return -1;
  }
  mcgpu_scan_options local;
  memset(&local, 0, sizeof local);
  memcpy(&local, caller_opt, std::min<size_t>(caller_opt->struct_size, sizeof local));
  local.struct_size = (unsigned int)sizeof local;
  const mcgpu_scan_options* opt = &local;
  for (int g = 0; g < n_ctx; ++g)
    if (!ctxs[g]) { mcgpu_set_last_error_("!!ERROR!! mcgpu_run_scan: null context"); return -1; }
  if (n_ctx > 1 && opt->shard == MCGPU_SHARD_PROJECTIONS) return run_scan_sharing_projections(ctxs, n_ctx, opt, report);
  // Routes of the per-projection tally sum between the devices (each is tried in its set-up phase, before anything is simulated or
  // written; every route gives the same output bytes):
  //   AUTO            the tally exchange (exchange.cpp: copy-engine pushes to the owner, one fused add); where the devices cannot reach
  //                   each other: projection sharding -- no traffic between the devices at all (SURVEY.md 8e's fallback mode)
  //   --reduce rccl   one ncclReduce per projection (reduce_rccl.cpp: north_star's collective, the reference's MPI_Reduce); where RCCL
  //                   cannot be set up: projection sharding
  // Round 6: AUTO no longer tries RCCL between the two.  The RCCL route has only ever run over a communicator of one
  // rank (no node); on a node without peer access it would stage 45 MB x (n - 1) per projection through host memory, where
  // projection sharding needs no traffic and writes the same bytes.  It is taken when asked for, and only then.
  const char* env_reduce = knob_str("MCGPU_REDUCE");
  // (asked for explicitly, the RCCL route also runs over ONE device -- a communicator of one rank: everything but the transport
  // between devices is then exercised on a one-GPU box, tests/test_gpu_dropin.py)
  const bool want_rccl = opt->reduce == MCGPU_REDUCE_RCCL || (opt->reduce == MCGPU_REDUCE_AUTO && env_reduce && !strcasecmp(env_reduce, "rccl"));
  auto say = [&](const char* what) {
    if (!opt->progress) return;
    // the reason, without the word the reference's log scanner takes for a failed run (cbctmc/mc/simulation.py:204)
    std::string why = mcgpu_last_error();
    for (size_t at = 0; at + 5 <= why.size(); ++at)
      if (strncasecmp(why.c_str() + at, "error", 5) == 0) why.replace(at, 5, "fault");
    for (size_t at; (at = why.find("!!")) != std::string::npos;) why.erase(at, 2);
    printf("       %s (%s)\n", what, why.c_str());
    fflush(stdout);
  };
  if (!want_rccl) {
    const int rc = run_scan_sharing_histories(ctxs, n_ctx, opt, report, false);
    if (rc != kExchangeUnavailable) return rc;
    // no peer access / no copy-engine path between these devices: the reference's split (histories of one projection on several
    // devices) needs a way to sum the tallies; whole projections per device need none and give the same files
    say("Tally exchange between the devices is not available: every device simulates whole projections instead (--reduce rccl asks for one RCCL reduction per projection)");
    return run_scan_sharing_projections(ctxs, n_ctx, opt, report);
  }
  const int rc = run_scan_sharing_histories(ctxs, n_ctx, opt, report, true);
  if (rc == 0 && opt->progress) { printf("       Detector tallies summed with one RCCL reduction (uint64, sum) per projection\n"); fflush(stdout); }
  if (rc != kRcclUnavailable) return rc;
  if (n_ctx == 1) {
    say("The RCCL reduction is not available: the device's own tally is the sum");
    return run_scan_sharing_histories(ctxs, n_ctx, opt, report, false);
  }
  say("The RCCL reduction that was asked for is not available: every device simulates whole projections instead");
  return run_scan_sharing_projections(ctxs, n_ctx, opt, report);
}

extern "C" int mcgpu_run_scan(mcgpu_ctx* ctx, const mcgpu_scan_options* opt, mcgpu_scan_report* report) {
  return mcgpu_run_scan_multi(&ctx, 1, opt, report);
}
