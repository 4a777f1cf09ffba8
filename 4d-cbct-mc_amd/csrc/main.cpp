// main.cpp -- drop-in executable `MC-GPU_v1.3.x <input.in>` (docker/mcgpu/MC-GPU_v1.3.cu:377-1214).
//
// Same command line, same input/output files and the same progress line on stdout
// ("<< Simulating Projection i of n >>", parsed by cbctmc/mc/simulation.py:200-219); the word "error"
// is only ever printed on failure (simulation.py:204 greps for it).  Options after the input file:
//   --mode fast|compat   kernel personality (default fast)
//   --gpus N             history-shard every projection over N devices of this node (default 1);
//                        per-device tallies are summed on the host (integers: order-independent)
//   --devices a,b,...    the same with an explicit device list (a device may appear twice: used by the tests to run the
//                        sharded path on a single-GPU box)
//   --no-output          skip the ASCII projection files (timing runs, or stacks only)
//   --stacks             also write projections_{total,unscattered,scattered}.mha next to the projection files
//                        (what cbctmc/mc/simulation.py:235-277 builds from the ASCII files afterwards)
//   --crop N             half-fan crop of the stacks (default 1024 when the detector has 1848 columns, else none)
//   --air FILE           air scan's projections_total.mha: also write projections_total_normalized.mha
// One GPU runs the pipelined scan driver (mcgpu_run_scan); several GPUs use the per-projection loop below.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcgpu_amd.h"

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  if (argc < 2) {
    printf("\n\n   !!read_input ERROR!! Input file name not given as an execution parameter!! Try again...\n\n");
    return 255;
  }
  int mode = MCGPU_MODE_FAST, ngpu = 1;
  bool write_out = true, stacks = false;
  int crop = -1;
  const char* air = nullptr;
  std::vector<int> device_list;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
      for (const char* s = argv[++i]; *s;) {
        device_list.push_back(atoi(s));
        while (*s && *s != ',') ++s;
        if (*s == ',') ++s;
      }
      continue;
    }
    if (!strcmp(argv[i], "--stacks")) { stacks = true; continue; }
    if (!strcmp(argv[i], "--crop") && i + 1 < argc) { crop = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--air") && i + 1 < argc) { air = argv[++i]; continue; }
    if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = !strcmp(argv[++i], "compat") ? MCGPU_MODE_COMPAT : MCGPU_MODE_FAST;
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) ngpu = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-output")) write_out = false;
  }
  if (!device_list.empty()) ngpu = (int)device_list.size();
  if (ngpu < 1) ngpu = 1;
  const double t_begin = now_s();
  printf("\n     *** MC CBCT projection engine for AMD MI355X (MC-GPU v1.3 file contract) ***\n\n    -- INITIALIZATION phase:\n");
  fflush(stdout);
  std::vector<mcgpu_ctx*> ctx(ngpu, nullptr);
  long long gpu_id = 0;
  for (int g = 0; g < ngpu; ++g) {
    // single GPU: the input file's GPU number; several: devices 0..N-1
    int dev = device_list.empty() ? g : device_list[g];
    if (ngpu == 1 && device_list.empty()) {
      mcgpu_ctx* probe = nullptr;
      if (mcgpu_create(argv[1], -1, &probe) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 254; }
      mcgpu_config_i64(probe, "gpu_id", &gpu_id);
      mcgpu_destroy(probe);
      dev = gpu_id > 0 ? (int)gpu_id : 0;
    }
    if (mcgpu_create(argv[1], dev, &ctx[g]) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 254; }
  }
  long long nproj = 1, hist = 0, seed = 0, tpb = 128, hpt = 150;
  mcgpu_config_i64(ctx[0], "num_projections", &nproj);
  mcgpu_config_i64(ctx[0], "total_histories", &hist);
  mcgpu_config_i64(ctx[0], "seed", &seed);
  mcgpu_config_i64(ctx[0], "threads_per_block", &tpb);
  mcgpu_config_i64(ctx[0], "histories_per_thread", &hpt);
  double d_angle = 0, a0 = 0, roi0 = 0, roi1 = 0;
  mcgpu_config_f64(ctx[0], "D_angle", &d_angle);
  mcgpu_config_f64(ctx[0], "initial_angle", &a0);
  mcgpu_config_f64(ctx[0], "angularROI_0", &roi0);
  mcgpu_config_f64(ctx[0], "angularROI_1", &roi1);
  size_t words = 0;
  mcgpu_image_words(ctx[0], &words);
  printf("\n    -- INITIALIZATION finished: elapsed time = %.3f s. \n\n\n    -- MONTE CARLO LOOP phase.\n\n", now_s() - t_begin);
  fflush(stdout);

  long long det_nx = 0, det_nz = 0;
  mcgpu_config_i64(ctx[0], "num_pixels_x", &det_nx);
  mcgpu_config_i64(ctx[0], "num_pixels_z", &det_nz);
  if (crop < 0) crop = det_nx == 1848 ? 1024 : 0;  // cbctmc/defaults.py:61-64
  const int crop_nx = (crop > 0 && crop < det_nx) ? crop : (int)det_nx;
  int blocks = 1, hpt_eff = (int)hpt;
  unsigned long long total = (unsigned long long)hist;
  if (mode == MCGPU_MODE_COMPAT) mcgpu_launch_shape((unsigned long long)hist, (int)tpb, (int)hpt, &blocks, &hpt_eff, &total);
  double t_mc = 0.0;
  if (ngpu == 1) {
    mcgpu_scan_options so;
    memset(&so, 0, sizeof so);
    so.mode = mode;
    so.progress = 1;
    so.crop_nx = crop_nx;
    so.write_ascii = write_out ? 1 : 0;
    so.write_stacks = stacks ? 1 : 0;
    so.air_stack = air;
    so.air_sigma_y = so.air_sigma_x = 10.0;
    so.pixel_spacing_x = so.pixel_spacing_y = 0.776;  // cbctmc/mc/projection.py:73
    mcgpu_scan_report sr;
    memset(&sr, 0, sizeof sr);
    if (mcgpu_run_scan(ctx[0], &so, &sr) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 253; }
    t_mc = sr.seconds_kernels;
    total = sr.histories_per_projection;
    printf("          *** SCAN PERFORMANCE REPORT ***\n              Projections:         %d\n              Simulated x rays:    %llu per projection\n"
           "              Kernel time [s]:     %.3f\n              Scan time [s]:       %.3f (output overlapped; %.3f s after the last kernel)\n"
           "              Speed [x-rays/s]:    %.2f\n\n",
           sr.projections, total, sr.seconds_kernels, sr.seconds_total, sr.seconds_after_last_kernel,
           sr.seconds_total > 0 ? (double)total * sr.projections / sr.seconds_total : 0.0);
  }
  std::vector<std::vector<uint64_t>> img(ngpu, std::vector<uint64_t>(ngpu > 1 ? words : 0));
  mcgpu_stack* stk[3] = {nullptr, nullptr, nullptr};
  std::vector<float> planes;
  std::string out_folder;
  if (ngpu > 1 && stacks) {
    char name[1024];
    mcgpu_projection_file_name(ctx[0], 0, name, sizeof name);
    out_folder = name;
    const size_t slash = out_folder.find_last_of('/');
    out_folder = slash == std::string::npos ? "." : out_folder.substr(0, slash);
    static const char* kNames[3] = {"projections_total.mha", "projections_unscattered.mha", "projections_scattered.mha"};
    for (int k = 0; k < 3; ++k)
      if (mcgpu_stack_create((out_folder + "/" + kNames[k]).c_str(), crop_nx, (int)det_nz, (int)nproj, 0.776, 0.776, &stk[k]) != 0) {
        printf("\n\n   %s\n\n", mcgpu_last_error());
        return 253;
      }
    planes.resize((size_t)3 * crop_nx * det_nz);
  }
  int cur_seed = (int)seed;
  const double RAD2DEG = 180.0 / 3.14159265358979323846;
  for (int p = 0; p < (int)nproj && ngpu > 1; ++p) {
    const double ang = a0 + p * d_angle;
    if (nproj != 1 && (ang < roi0 || ang > roi1)) {
      printf("         << Skipping projection #%d of %d >> Angle %f degrees: outside angular region of interest.\n", p + 1, (int)nproj, ang * RAD2DEG);
      continue;
    }
    if (nproj != 1) printf("\n\n\n   << Simulating Projection %d of %d >> Angle: %lf degrees.\n\n\n", p + 1, (int)nproj, ang * RAD2DEG);
    fflush(stdout);
    const double t0 = now_s();
    // units to shard: batches (compat) or histories (fast)
    const unsigned long long units = mode == MCGPU_MODE_COMPAT ? (unsigned long long)blocks * (unsigned long long)tpb : total;
    std::vector<int> rc(ngpu, 0);
    std::vector<std::string> err(ngpu);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g)
      th.emplace_back([&, g]() {
        const unsigned long long lo = units * g / ngpu, hi = units * (g + 1) / ngpu;
        rc[g] = mcgpu_run_projection(ctx[g], p, mode, cur_seed, lo, hi - lo, hpt_eff, img[g].data(), nullptr, nullptr);
        if (rc[g]) err[g] = mcgpu_last_error();
      });
    for (auto& t : th) t.join();
    for (int g = 0; g < ngpu; ++g)
      if (rc[g]) { printf("\n\n   %s\n\n", err[g].c_str()); return 253; }
    {  // integer sum of the per-device tallies (the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019), banded over host threads
      const int T = 8;
      std::vector<std::thread> add;
      for (int t = 0; t < T; ++t)
        add.emplace_back([&, t]() {
          const size_t lo = words * t / T, hi = words * (t + 1) / T;
          for (int g = 1; g < ngpu; ++g) {
            const uint64_t* src = img[g].data();
            uint64_t* dst = img[0].data();
            for (size_t i = lo; i < hi; ++i) dst[i] += src[i];
          }
        });
      for (auto& t : add) t.join();
    }
    const double dt = now_s() - t0;
    t_mc += dt;
    printf("          *** IMAGE TALLY PERFORMANCE REPORT ***\n              CT projection %d of %d\n              Simulated x rays:    %llu\n"
           "              Simulation time [s]: %.2f\n              Speed [x-rays/s]:    %.2f\n\n", p + 1, (int)nproj, total, dt, dt > 0 ? total / dt : 0.0);
    if (write_out && mcgpu_write_projection(ctx[0], p, img[0].data(), total, dt, nullptr) != 0) {
      printf("\n\n   %s\n\n", mcgpu_last_error());
      return 253;
    }
    if (stk[0]) {
      int src = mcgpu_finalize_projection_host(ctx[0], img[0].data(), total, crop_nx, planes.data());
      for (int k = 0; k < 3 && src == 0; ++k) src = mcgpu_stack_append(stk[k], planes.data() + (size_t)k * crop_nx * det_nz);
      if (src != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 253; }
    }
    // next projection gets a disjoint stream set (update_seed_PRNG, MC-GPU_v1.3.cu:869)
    if (mode == MCGPU_MODE_COMPAT) cur_seed = mcgpu_advance_seed(1, total, cur_seed);
    fflush(stdout);
  }
  if (stk[0]) {
    for (int k = 0; k < 3; ++k)
      if (mcgpu_stack_finish(stk[k], 1, nullptr) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 253; }
    if (air && mcgpu_normalize_stack((out_folder + "/projections_total.mha").c_str(), air, 10.0, 10.0,
                                     (out_folder + "/projections_total_normalized.mha").c_str(), 0.776, 0.776) != 0) {
      printf("\n\n   %s\n\n", mcgpu_last_error());
      return 253;
    }
  }
  // dose tallies accumulate over all projections (MC-GPU_v1.3.cu:1062-1165): sum the devices, report once
  int dose_flags = 0;
  size_t roi_voxels = 0;
  mcgpu_dose_info(ctx[0], &dose_flags, nullptr, &roi_voxels);
  if (dose_flags != 0) {
    std::vector<uint64_t> vox(dose_flags & 2 ? 2 * roi_voxels : 0), mat(dose_flags & 1 ? 50 : 0), tmp_v(vox.size()), tmp_m(mat.size());
    for (int g = 0; g < ngpu; ++g) {
      if (mcgpu_dose_read(ctx[g], vox.empty() ? nullptr : tmp_v.data(), mat.empty() ? nullptr : tmp_m.data()) != 0) {
        printf("\n\n   %s\n\n", mcgpu_last_error());
        return 253;
      }
      for (size_t i = 0; i < vox.size(); ++i) vox[i] += tmp_v[i];
      for (size_t i = 0; i < mat.size(); ++i) mat[i] += tmp_m[i];
    }
    if (mcgpu_write_dose_report(ctx[0], vox.empty() ? nullptr : vox.data(), mat.empty() ? nullptr : mat.data(), total, now_s() - t_begin, nullptr, 0) != 0) {
      printf("\n\n   %s\n\n", mcgpu_last_error());
      return 253;
    }
  }
  for (auto* c : ctx) mcgpu_destroy(c);
  const double t_all = now_s() - t_begin;
  printf("\n\n\n    -- SIMULATION FINISHED!\n\n          >>> Execution time including initialization, transport and report: %.3f s.\n"
         "          >>> Time spent in the Monte Carlo transport only: %.3f s.\n          >>> Total number of simulated x rays:  %llu\n",
         t_all, t_mc, total * (unsigned long long)nproj);
  return 0;
}
