// main.cpp -- drop-in executable `MC-GPU_v1.3.x <input.in>` (docker/mcgpu/MC-GPU_v1.3.cu:377-1214).
//
// Same command line, same input/output files and the same progress line on stdout
// ("<< Simulating Projection i of n >>", parsed by cbctmc/mc/simulation.py:200-219); the word "error"
// is only ever printed on failure (simulation.py:204 greps for it).  Options after the input file:
//   --mode fast|compat   kernel personality (default fast)
//   --gpus N             history-shard every projection over devices 0..N-1 of this node (default 1: the input file's
//                        GPU number); the per-device tallies are summed through the tally exchange (exchange.cpp: copy-engine
//                        pushes to the projection's owner device, one fused add; integers: order-independent)
//   --devices a,b,...    the same with an explicit device list (a device may appear twice: used by the tests to run the
//                        sharded path on a single-GPU box)
//   --shard histories|projections   with several devices: share every projection's histories (default: the reference's split,
//                        tallies summed through the exchange) or give every device whole projections (no traffic between the
//                        devices at all: the fallback for nodes without working peer access; same output bytes)
//   --reduce exchange|rccl   with several devices sharing histories: sum the per-device tallies through the tally exchange (default; falls
//                        back to RCCL, then to projection sharding, where the devices cannot reach each other) or with ONE
//                        ncclReduce(uint64, sum) per projection (the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019; RCCL is opened with dlopen)
//   --no-output          skip the ASCII projection files (timing runs, or stacks only)
//   --stacks             also write projections_{total,unscattered,scattered}.mha next to the projection files
//                        (what cbctmc/mc/simulation.py:235-277 builds from the ASCII files afterwards)
//   --crop N             half-fan crop of the stacks (default 1024 when the detector has 1848 columns, else none)
//   --air FILE           air scan's projections_total.mha: also write projections_total_normalized.mha
// Everything runs through the pipelined scan driver (mcgpu_run_scan_multi, scan.cpp).
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
  if (argc >= 2 && !strcmp(argv[1], "--knobs")) {  // the environment knobs of the engine (csrc/knobs.cpp), then exit
    std::vector<char> t(mcgpu_knob_table(nullptr, 0));
    mcgpu_knob_table(t.data(), t.size());
    printf("# name\ttype (i int, f float, b switch, s string)\tscope (K kernel variant / schedule, H host pipeline, T test hook, P Python side)\tdefault\tcurrent\twhat\n%s", t.data());
    return 0;
  }
  int mode = MCGPU_MODE_FAST, ngpu = 1, shard = MCGPU_SHARD_HISTORIES, reduce = MCGPU_REDUCE_AUTO;
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
    } else if (!strcmp(argv[i], "--stacks")) stacks = true;
    else if (!strcmp(argv[i], "--crop") && i + 1 < argc) crop = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--air") && i + 1 < argc) air = argv[++i];
    else if (!strcmp(argv[i], "--mode") && i + 1 < argc) { ++i; mode = !strcmp(argv[i], "compat") ? MCGPU_MODE_COMPAT : (!strcmp(argv[i], "fast64") ? MCGPU_MODE_FAST_F64 : MCGPU_MODE_FAST); }
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) ngpu = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--shard") && i + 1 < argc) shard = !strcmp(argv[++i], "projections") ? MCGPU_SHARD_PROJECTIONS : MCGPU_SHARD_HISTORIES;
    else if (!strcmp(argv[i], "--reduce") && i + 1 < argc) reduce = !strcmp(argv[++i], "rccl") ? MCGPU_REDUCE_RCCL : MCGPU_REDUCE_AUTO;
    else if (!strcmp(argv[i], "--no-output")) write_out = false;
  }
  if (!device_list.empty()) ngpu = (int)device_list.size();
  if (ngpu < 1) ngpu = 1;
  const double t_begin = now_s();
  printf("\n     *** MC CBCT projection engine for AMD MI355X (MC-GPU v1.3 file contract) ***\n\n    -- INITIALIZATION phase:\n");
  fflush(stdout);
  // The input, the voxel file and the material files are parsed ONCE (host-only context); every device gets a clone, and
  // the clones are built side by side (the reference parses everything in every MPI rank, MC-GPU_v1.3.cu:377-640).
  mcgpu_ctx* parsed = nullptr;
  if (mcgpu_create(argv[1], -1, &parsed) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 254; }
  std::vector<int> devs(ngpu, 0);
  for (int g = 0; g < ngpu; ++g) devs[g] = device_list.empty() ? g : device_list[g];
  if (ngpu == 1 && device_list.empty()) {  // single GPU: the input file's GPU number
    long long gpu_id = 0;
    mcgpu_config_i64(parsed, "gpu_id", &gpu_id);
    devs[0] = gpu_id > 0 ? (int)gpu_id : 0;
  }
  std::vector<mcgpu_ctx*> ctx(ngpu, nullptr);
  {
    std::vector<std::string> err(ngpu);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g)
      th.emplace_back([&, g] { if (mcgpu_clone(parsed, devs[g], &ctx[g]) != 0) err[g] = mcgpu_last_error(); });  // the last error is per thread
    for (auto& t : th) t.join();
    mcgpu_destroy(parsed);
    for (int g = 0; g < ngpu; ++g)
      if (!ctx[g]) { printf("\n\n   %s\n\n", err[g].empty() ? "!!ERROR!! device context could not be created" : err[g].c_str()); return 254; }
  }
  long long det_nx = 0;
  mcgpu_config_i64(ctx[0], "num_pixels_x", &det_nx);
  if (crop < 0) crop = det_nx == 1848 ? 1024 : 0;  // cbctmc/defaults.py:61-64
  printf("\n    -- INITIALIZATION finished: elapsed time = %.3f s. \n\n\n    -- MONTE CARLO LOOP phase.\n\n", now_s() - t_begin);
  fflush(stdout);

  mcgpu_scan_options so;
  memset(&so, 0, sizeof so);
  so.struct_size = (unsigned int)sizeof so;
  so.mode = mode;
  so.progress = 1;
  so.shard = shard;
  so.reduce = reduce;
  so.crop_nx = (crop > 0 && crop < det_nx) ? crop : (int)det_nx;
  so.write_ascii = write_out ? 1 : 0;
  so.write_stacks = stacks ? 1 : 0;
  so.air_stack = air;
  so.air_sigma_y = so.air_sigma_x = 10.0;               // cbctmc/mc/simulation.py:239
  so.pixel_spacing_x = so.pixel_spacing_y = 0.776;      // cbctmc/mc/projection.py:73
  mcgpu_scan_report sr;
  memset(&sr, 0, sizeof sr);
  if (mcgpu_run_scan_multi(ctx.data(), ngpu, &so, &sr) != 0) { printf("\n\n   %s\n\n", mcgpu_last_error()); return 253; }
  const unsigned long long total = sr.histories_per_projection;
  printf("          *** SCAN PERFORMANCE REPORT ***\n              Devices:             %d\n              Projections:         %d\n"
         "              Simulated x rays:    %llu per projection\n              Kernel time [s]:     %.3f\n"
         "              Scan time [s]:       %.3f (output overlapped; %.3f s after the last kernel)\n              Speed [x-rays/s]:    %.2f\n\n",
         ngpu, sr.projections, total, sr.seconds_kernels, sr.seconds_total, sr.seconds_after_last_kernel,
         sr.seconds_total > 0 ? (double)total * sr.projections / sr.seconds_total : 0.0);

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
         t_all, sr.seconds_kernels, total * (unsigned long long)sr.projections);
  return 0;
}
