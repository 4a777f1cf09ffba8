/*
 * mcgpu_amd.h -- C ABI of the MI355X-native Monte Carlo CBCT projection engine.
 *
 * The reference (IPMI-ICNS-UKE/4d-cbct-mc) exposes NO in-process interface for this path: its
 * boundary is process + files (`mpirun -n <ngpu> MC-GPU_v1.3.x input.in`, cbctmc/mc/simulation.py:187-198;
 * SURVEY.md 8b).  This header factors that executable's main() (docker/mcgpu/MC-GPU_v1.3.cu:377-1214)
 * into entry points with plain pointers and sizes, so the same engine can be driven by
 *   - the drop-in executable `MC-GPU_v1.3.x` (4d-cbct-mc_amd/csrc/main.cpp),
 *   - the Python mirror of cbctmc.mc (ctypes, 4d-cbct-mc_amd/engine.py),
 *   - tests and bench.py.
 * Each entry point cites the reference code it replaces.  All functions return 0 on success or a
 * negative error code (the reference's exit codes: -1 input/GPU, -2 parse/alloc, -3 output);
 * mcgpu_last_error() gives the message (it always contains "ERROR", which is what
 * cbctmc/mc/simulation.py:204 greps the engine log for).  No exceptions cross the ABI.
 * Ownership: the context owns host tables and device tables; callers own image buffers.
 */
#ifndef MCGPU_AMD_H_
#define MCGPU_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcgpu_ctx mcgpu_ctx;

/* Kernel personalities.
 * FAST   : production path -- counter-based per-history RNG streams (Philox4x32-7 seeding a
 *          multiply-with-carry lane generator), gfx950 native transcendental instructions, exactly
 *          n_histories histories; statistically equivalent to the reference (3-sigma per pixel).
 * COMPAT : RANECU leap-frog streams (batch <-> thread mapping of MC-GPU_kernel_v1.3.cu:198,841-894),
 *          the reference CPU-build arithmetic and the portable math of oracle/mcgpu_oracle.c;
 *          integer tallies are bit-identical to the CPU oracle. */
#define MCGPU_MODE_FAST 0
#define MCGPU_MODE_COMPAT 1
/* FAST with scheduler statistics (diagnostic build of the same kernel; see mcgpu_scheduler_stats). Not for timing. */
#define MCGPU_MODE_FAST_STATS 2
/* FAST with the three sub-steps the reference computes in double precision -- rotate_double (MC-GPU_kernel_v1.3.cu:1103-1148),
 * GRAa (:1181-1246), GCOa's cdt1 / costh chain (:1329-1331, :1372, :1427) -- in double here too (32-bit deviates where the
 * reference calls ranecu_double): the production kernel at the reference's arithmetic.  Same scheduling, same streams. */
#define MCGPU_MODE_FAST_F64 3

int mcgpu_abi_version(void);
/* The engine's environment knobs as text: one line per knob, tab-separated {name, type (i/f/b/s), scope (K kernel variant or
 * schedule, H host pipeline, T test hook, P Python side), default, current value, description}.  Returns the bytes the whole
 * table needs including the NUL (call with cap = 0 to size a buffer).  No counterpart in the reference (MC-GPU_v1.3.cu reads
 * no environment); `MC-GPU_v1.3.x --knobs` prints it.  A variable MCGPU_* that is not in the table is reported once per
 * process on stdout when a context is created. */
size_t mcgpu_knob_table(char *buf, size_t cap);
const char *mcgpu_last_error(void);

/* read_input + init_energy_spectrum + set_CT_trajectory + load_voxels + load_material
 * (MC-GPU_v1.3.cu:490-562) and, when device_id >= 0, init_CUDA_device (:2454-2724): select the
 * device and upload all tables.  device_id < 0 builds the host model only (no HIP call is made). */
int mcgpu_create(const char *input_path, int device_id, mcgpu_ctx **out);
void mcgpu_destroy(mcgpu_ctx *ctx);
/* A second context of the same simulation on another device (device_id < 0: host only) without parsing the input, the
 * voxel file and the material files again: what each further rank of the reference's `mpirun -n N` repeats in full
 * (MC-GPU_v1.3.cu:377-640: every MPI process runs read_input / load_voxels / load_material).  The clone is independent
 * of `src` afterwards. */
int mcgpu_clone(const mcgpu_ctx *src, int device_id, mcgpu_ctx **out);

/* Scalars parsed from the input file (MC-GPU_v1.3.cu:1280-1615).  Keys: "total_histories", "seed",
 * "gpu_id", "threads_per_block", "histories_per_thread", "num_projections", "enable_specific_angles",
 * "num_voxels_x|y|z", "num_pixels_x|z", "num_materials_used", "num_energy_values", "palette_size",
 * "volume_bytes_device"; of the device model (diagnostic): "volume_kind", "brick_shift", "brick_count", "bricks_mixed",
 * "bricks_exterior", "exterior_cylinder", "tile_records", "sub_brick_table", "fast_scheduler", "segment_loop", "tiles_in_mixed_bricks", "sigma_bracket_shift", "lds_bytes_fast",
 * "lds_bytes_compat", "blocks_per_cu", "num_cus", "device_id". */
int mcgpu_config_i64(const mcgpu_ctx *ctx, const char *key, long long *value);
/* Keys: "D_angle", "initial_angle", "angularROI_0", "angularROI_1", "SRotAxisD", "vertical_translation",
 * "mean_energy_spectrum", "e0", "ide". */
int mcgpu_config_f64(const mcgpu_ctx *ctx, const char *key, double *value);

/* Host tables in the reference's own layouts (MC-GPU_v1.3.h:155-264), for cross-checks against the
 * reference/oracle: "source_data" (80 B x nproj), "detector_data" (100 B x nproj), "voxel_mat_dens"
 * (float2 x nvox, built on demand), "mfp_woodcock", "mfp_a", "mfp_b", "xco","pco","aco","bco","pmax",
 * "itlco","ituco","fco","uico","fj0","noscco","espc","espc_cutoff","espc_alias","density_max",
 * "density_nominal","voxel_size","inv_voxel_size","size_bbox".  The pointer stays valid until destroy. */
int mcgpu_host_table(mcgpu_ctx *ctx, const char *name, const void **data, size_t *bytes);

/* Output file name of projection p: "<base>_%010.6fdeg" with the float32 angle (MC-GPU_v1.3.cu:2787-2803). */
int mcgpu_projection_file_name(const mcgpu_ctx *ctx, int p, char *buf, size_t buf_bytes);

/* Number of uint64 tally words per projection: 4 * Nx * Nz (MC-GPU_v1.3.cu:1848-1849). */
int mcgpu_image_words(const mcgpu_ctx *ctx, size_t *words);

/* Launch sizing of the reference (MC-GPU_v1.3.cu:823-841): blocks of `threads` threads, `hpt`
 * histories per thread (raised when blocks would exceed 65535); total = blocks*threads*hpt. */
int mcgpu_launch_shape(unsigned long long histories, int threads_per_block, int histories_per_thread, int *blocks,
                       int *hpt_out, unsigned long long *total_histories);

/* update_seed_PRNG (MC-GPU_v1.3.cu:3456-3485): seed for the next projection. */
int mcgpu_advance_seed(int batch_number, unsigned long long total_histories, int seed);

/* track_particles<<<>>> for projection p (MC-GPU_v1.3.cu:861; kernel MC-GPU_kernel_v1.3.cu:120-384),
 * asynchronous on `hip_stream` (a hipStream_t, NULL = default stream), ADDING into the caller's device
 * buffer image_dev[4*Nx*Nz] (uint64).  History sharding: the launch simulates
 *   FAST  : history ids [first, first+count)            (count = n_histories for a single GPU)
 *   COMPAT: batches     [first, first+count) of `hpt` histories each (batch b uses RANECU stream b)
 * so ranks given disjoint ranges produce images whose sum equals the single-GPU image exactly.
 * `seed` is the RNG seed for this projection; `hpt` is ignored in FAST mode. */
int mcgpu_launch_projection(mcgpu_ctx *ctx, int p, int mode, int seed, unsigned long long first, unsigned long long count,
                            int hpt, void *image_dev, void *hip_stream);
/* Milliseconds between the HIP events recorded around the most recent launch on its stream
 * (synchronises on the stop event). */
int mcgpu_last_kernel_ms(mcgpu_ctx *ctx, float *ms);
/* Scheduler statistics accumulated by MCGPU_MODE_FAST_STATS launches: out8 = {wave loop iterations, sum of flying
 * lanes, Compton rounds, Compton lanes, Rayleigh rounds, Rayleigh lanes, tally+source rounds, their lanes}. */
int mcgpu_scheduler_stats(mcgpu_ctx *ctx, unsigned long long *out8, int reset);
/* Batching thresholds of the FAST kernel, in lanes of a wave64: pending Compton / Rayleigh / tally+source histories that
 * trigger a batch, the number of lanes able to fly below which every well-populated kind is served, and how many lanes park
 * between two scheduling points.  Results never depend on them (per-history RNG streams, integer tallies); speed does, by a
 * few percent between geometries.  mcgpu_run_scan picks among a few presets with short throw-away launches unless
 * MCGPU_THRESH_* / MCGPU_FLYABLE_LOW / MCGPU_SWAP_BATCH are set in the environment (those always win). */
int mcgpu_set_fast_schedule(mcgpu_ctx *ctx, int thresh_compton, int thresh_rayleigh, int thresh_new, int flyable_low, int swap_batch);
/* The tuning knobs of the environment (MCGPU_THRESH_*, MCGPU_FLYABLE_LOW, MCGPU_SWAP_BATCH, MCGPU_SLOT_TRADE,
 * MCGPU_HOLD_Q, MCGPU_EXTERIOR_MODE, MCGPU_BLOCKS_PER_CU, MCGPU_GRID_SPARE_PERCENT, MCGPU_COMPAT_THRESH_*) are read ONCE, when
 * the context's device model is built; mcgpu_launch_projection never reads the environment and never synchronises.  Tuning
 * tools that change them in a live process call this to read them again (drains the device, re-uploads the parameter block).
 * The reference has no counterpart: its launch shape is fixed by the input file (MC-GPU_v1.3.cu:823-841). */
int mcgpu_reload_env_knobs(mcgpu_ctx *ctx);
/* The same with the full counter set (up to 16): 8 = scheduling points, 9/10 = register<->LDS-slot exchange rounds /
 * lanes that bring a flying history in, 11 = scheduling points in drain mode. */
int mcgpu_scheduler_stats_ex(mcgpu_ctx *ctx, unsigned long long *out, int capacity, int reset);
/* Convenience for callers without a HIP binding of their own (ctypes hosts, tests): copy `bytes` from device memory of the
 * context's device to host memory, ordered behind everything enqueued on `hip_stream` so far; returns when the data is there. */
int mcgpu_copy_to_host(mcgpu_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes, void *hip_stream);

/* hipMemsetAsync of an image buffer (init_image_array_GPU, MC-GPU_kernel_v1.3.cu:56-72). */
int mcgpu_clear_image(mcgpu_ctx *ctx, void *image_dev, void *hip_stream);

/* Convenience, synchronous: allocate+zero a device image, launch, wait, copy to image_host[4*Nx*Nz]
 * (MC-GPU_v1.3.cu:861-907).  kernel_seconds / histories_done may be NULL. */
int mcgpu_run_projection(mcgpu_ctx *ctx, int p, int mode, int seed, unsigned long long first, unsigned long long count, int hpt,
                         uint64_t *image_host, double *kernel_seconds, unsigned long long *histories_done);

/* report_image (MC-GPU_v1.3.cu:2783-2953): write the ASCII projection file (file_name NULL = the
 * reference's name for projection p).  Values = image * (1/100) * inv_px_X * inv_px_Z / total_histories. */
int mcgpu_write_projection(mcgpu_ctx *ctx, int p, const uint64_t *image_host, unsigned long long total_histories, double seconds,
                           const char *file_name);

/* report_image with the 1.4 M data lines formatted ON THE DEVICE (MC-GPU_v1.3.cu:2860-2904: four "%.8lf" numbers per pixel,
 * a blank line per detector row; exact decimal conversion in integer arithmetic, byte-identical to mcgpu_write_projection).
 * mcgpu_format_projection is asynchronous on `hip_stream`: it reads the device tally `image_dev` (before it is cleared) and
 * fills one of MCGPU_ASCII_SLOTS internal slots and records an event behind it; the host side
 * mcgpu_write_formatted_projection waits for that event, downloads the text with a copy engine and writes header + text +
 * footer.  A tally with a value outside the formatter's range (>= 1e11 eV/cm^2 per history; unreachable for photon physics)
 * makes it return -3 -- there is no host fallback on this route (the tally has been cleared by then).  The scan driver
 * cycles through the slots with one writer thread each, so the files of projections i - 2 .. i are written side by side
 * while i + 1 is tracked and formatted.  Different slots may be used from different threads at the same time. */
#define MCGPU_ASCII_SLOTS 3
int mcgpu_format_projection(mcgpu_ctx *ctx, const void *image_dev, unsigned long long total_histories, int slot, void *hip_stream);
int mcgpu_write_formatted_projection(mcgpu_ctx *ctx, int p, int slot, unsigned long long total_histories, double seconds,
                                     const char *file_name);

/* Dose tallies (tally_materials_dose MC-GPU_kernel_v1.3.cu:1547-1563, tally_voxel_energy_deposition :418-443; enabled by
 * SECTION DOSE DEPOSITION of the input file, MC-GPU_v1.3.cu:1619-1709).  The context owns device buffers that every
 * launch adds into (all projections accumulate, as in the reference).
 *   flags: bit 0 = material dose tally, bit 1 = voxel dose tally; roi6 = 0-based inclusive xmin,xmax,ymin,ymax,zmin,zmax;
 *   voxels: uint64 pairs {Edep*100, Edep^2} per ROI voxel (x fastest); materials: uint64 pairs x 25 material numbers. */
int mcgpu_dose_info(const mcgpu_ctx *ctx, int *flags, int roi6[6], size_t *roi_voxels);
int mcgpu_dose_read(mcgpu_ctx *ctx, uint64_t *voxels_out /* 2*roi_voxels or NULL */, uint64_t *materials_out /* 50 or NULL */);
int mcgpu_dose_clear(mcgpu_ctx *ctx);
/* report_voxels_dose (MC-GPU_v1.3.cu:2976-3199: ASCII z-plane file + <file>.raw + <file>_2sigma.raw) when `voxels` is
 * given, report_materials_dose (:3214-3262) when `materials` is given.  The text the reference prints to stdout goes to
 * `log` (NUL-terminated, truncated to log_bytes) or, when log is NULL, to stdout. */
int mcgpu_write_dose_report(mcgpu_ctx *ctx, const uint64_t *voxels, const uint64_t *materials, unsigned long long histories_per_projection,
                            double seconds, char *log, size_t log_bytes);

/* ---- projection post-processing and RTK-ready stacks (cbctmc/mc/projection.py:36-169, simulation.py:235-277) ----
 * The reference's Python reads every ASCII projection back (np.loadtxt -> float32), flips z, crops the half-fan columns
 * and stacks {total, unscattered, scattered} into MetaImage files.  These entry points produce the same float32 numbers
 * from the integer tallies directly.  planes = float[3][Nz][crop_nx] (total, unscattered, scattered); crop_nx <= 0 or
 * >= Nx keeps the full width. */
int mcgpu_finalize_projection(mcgpu_ctx *ctx, void *image_dev, unsigned long long total_histories, int crop_nx, void *planes_dev,
                              int clear_image, void *hip_stream);
int mcgpu_finalize_projection_host(const mcgpu_ctx *ctx, const uint64_t *image_host, unsigned long long total_histories, int crop_nx,
                                   float *planes_host);
/* MetaImage float32 stack written plane by plane (projections_to_itk, projection.py:118-166: spacing (sx, sy, 1), origin
 * (-nx*sx/2, -ny*sy/2, 0)).  finish(replace_zeros != 0) applies np.where(stack == 0, stack[stack > 0].min(), stack). */
typedef struct mcgpu_stack mcgpu_stack;
int mcgpu_stack_create(const char *path, int nx, int ny, int nslices, double spacing_x, double spacing_y, mcgpu_stack **out);
int mcgpu_stack_append(mcgpu_stack *stack, const float *plane);
int mcgpu_stack_write_slice(mcgpu_stack *stack, int slice, const float *plane); /* any order, each slice once; not mixed with append */
int mcgpu_stack_finish(mcgpu_stack *stack, int replace_zeros, float *replacement_value);
int mcgpu_stack_read(const char *path, int dims3[3], float *data /* NULL: dims only */, size_t capacity_elements);
/* normalize_projections (projection.py:96-115): out = log(gaussian_filter(air, (sigma_y, sigma_x)) / total), scipy's
 * gaussian_filter semantics (truncate 4, reflect); sigma <= 0 skips the filter. */
int mcgpu_normalize_stack(const char *total_stack, const char *air_stack, double sigma_y, double sigma_x, const char *out_stack,
                          double spacing_x, double spacing_y);

/* Whole-scan driver on the context's GPU: the projection loop of main() (MC-GPU_v1.3.cu:667-1056) as a pipeline --
 * track -> finalize (+ clear) on the device, double-buffered pinned copies, a writer thread for the files -- so the GPU
 * never waits for output.  Zero-initialise the struct, THEN set struct_size = sizeof(mcgpu_scan_options); every other field has a
 * usable default.  struct_size stays the first field for good: inserting anything before it would be a deliberate ABI break. */
typedef struct mcgpu_scan_options {
  /* sizeof(mcgpu_scan_options) as the CALLER was compiled: fields the caller's header did not have yet read as zero (their
   * defaults), instead of whatever follows the shorter struct in memory.  0 is refused. */
  unsigned int struct_size;
  int mode;                                     /* MCGPU_MODE_FAST (default) or MCGPU_MODE_COMPAT */
  int first_projection, num_projections;        /* num_projections 0 = all remaining */
  unsigned long long histories_per_projection;  /* 0 = the input file's value */
  int crop_nx;                                  /* half-fan crop of the stacks (reference default 1024); 0 = full width */
  int write_ascii;                              /* the reference's per-projection ASCII files */
  int write_stacks;                             /* projections_{total,unscattered,scattered}.mha in output_folder */
  const char *output_folder;                    /* NULL = folder of the input file's output base name */
  const char *air_stack;                        /* air scan's projections_total.mha: also write projections_total_normalized.mha */
  double air_sigma_y, air_sigma_x;              /* gaussian denoising of the air projection (reference default 10, 10) */
  double pixel_spacing_x, pixel_spacing_y;      /* MetaImage spacing [mm]; 0 = detector pixel size of the input file */
  /* 4-D scans: three caller-owned stacks {total, unscattered, scattered} shared by several scans; projection i of THIS
   * scan (i = 0 .. num_projections-1) is written to slice slice_of_projection[i].  The caller finishes the stacks. */
  mcgpu_stack **shared_stacks;
  const int *slice_of_projection;
  int progress;                                 /* print the reference's "<< Simulating Projection i of n >>" lines to stdout */
  /* mcgpu_run_scan_multi: how the work is split over the contexts.  MCGPU_SHARD_HISTORIES (0, default): the reference's split
   * (every projection's histories over the devices, tallies summed; MC-GPU_v1.3.cu:728-731, :1019).  MCGPU_SHARD_PROJECTIONS:
   * context g simulates ALL histories of the simulated projections number g, g + n, g + 2n, ... and nothing crosses between the
   * devices (SURVEY.md 8e's fallback: no exchange, no collective); outputs are identical to a one-device scan. */
  int shard;
  /* simulate only the simulated projections number phase, phase + stride, ... of the range (0, 0 = all): what
   * MCGPU_SHARD_PROJECTIONS hands to each context; usable directly by a host that runs one mcgpu_run_scan per device */
  int projection_stride, projection_phase;
  /* mcgpu_run_scan_multi with MCGPU_SHARD_HISTORIES: how the per-device tallies of a projection are summed (the reference's
   * MPI_Reduce, MC-GPU_v1.3.cu:1006-1024).  MCGPU_REDUCE_AUTO (0): the tally exchange; where the devices cannot reach each other,
   * one RCCL reduction per projection; where that is not available either, projection sharding -- unless the environment says
   * MCGPU_REDUCE=rccl.  MCGPU_REDUCE_RCCL: one ncclReduce(uint64, sum, root = the projection's owner) per projection on a stream of
   * its own beside the next projection's kernel (then projection sharding if RCCL cannot be set up).  Same output bytes on every route. */
  int reduce;
} mcgpu_scan_options;
#define MCGPU_SHARD_HISTORIES 0
#define MCGPU_SHARD_PROJECTIONS 1
#define MCGPU_REDUCE_AUTO 0
#define MCGPU_REDUCE_RCCL 1
typedef struct mcgpu_scan_report {
  int projections;
  unsigned long long histories_per_projection;
  double seconds_total, seconds_kernels, seconds_after_last_kernel;
  float zero_replacement[3];
  double seconds_writer;                        /* busy time of the output thread (overlapped with tracking) */
  double kernel_ms_min, kernel_ms_max;          /* fastest / slowest projection of the scan (the slowest device's launch each): kernels vary with the angle */
} mcgpu_scan_report;
int mcgpu_run_scan(mcgpu_ctx *ctx, const mcgpu_scan_options *options, mcgpu_scan_report *report);
/* The same over several devices of one node (contexts created from the same input file, one per device): every
 * projection's histories are sharded over the contexts (the reference's `mpirun -n N`, MC-GPU_v1.3.cu:728-731,823-841) and the
 * per-device tallies are summed through the tally exchange below (the MPI_Reduce of :1019): every projection has an owner
 * device (projection mod devices; MCGPU_EXCHANGE_POLICY=0: always the first), the others push their tally to it with a copy
 * engine beside their next kernel, the owner adds them in one pass, finalizes, formats and downloads the projection; the
 * writer thread takes the results in projection order.  No device waits for the sum.  Dose tallies stay per context (sum
 * them with mcgpu_dose_read). */
int mcgpu_run_scan_multi(mcgpu_ctx *const *ctxs, int n_ctx, const mcgpu_scan_options *options, mcgpu_scan_report *report);

/* ---- The tally exchange between the GPUs of one node: the sum of the per-rank detector tallies that the reference does with
 * a device-to-host copy and a blocking MPI_Reduce to rank 0 per projection (MC-GPU_v1.3.cu:1006-1024).  One mcgpu_exchange
 * per rank -- a process with its own GPU (bench.py, one process per GPU) or a context of one process
 * (mcgpu_run_scan_multi) -- all built over one small host region `shared` of mcgpu_exchange_shared_bytes(world) ZEROED bytes
 * that every rank sees (plain memory inside one process, a mapped /dev/shm file between processes).  Every step
 * (projection) has an owner rank: policy 0 = rank 0, MCGPU_EXCHANGE_ROTATE = step mod world.  Per step, on the rank's
 * tracking stream:   begin(step) -> tally buffer (zeroed) | mcgpu_launch_projection into it | submit(step) |
 * collect(step - 1).  submit: a rank that does not own the step pushes its tally into the owner's landing buffer with a COPY
 * ENGINE (no kernel may run beside the persistent tracking grid), overlapped with the next projection's tracking.
 * collect: the owner adds the landed tallies to its own in one fused pass behind its next kernel and gets the complete
 * tally (valid until begin(step + 2)); other ranks get NULL.  Integer sums: the result equals the single-GPU tally bit for
 * bit.  Between processes ranks swap "cards" (IPC memory and event handles) once: card() on every rank, an all-gather by
 * whatever means the host has (its process-group library, MPI, a file), connect() to every peer.  A rank that waits for a peer
 * that has gone gets an error after 120 s, never a hang. */
typedef struct mcgpu_exchange mcgpu_exchange;
#define MCGPU_EXCHANGE_ROTATE 1 /* policy bit: owner of step s is rank s % world (default: rank 0, the reference's root) */
#define MCGPU_EXCHANGE_LOCAL 2  /* policy bit: every rank is a context of THIS process (plain events, connect_local) */
size_t mcgpu_exchange_shared_bytes(int world);
size_t mcgpu_exchange_card_bytes(int world);
int mcgpu_exchange_create(int device_id, int rank, int world, size_t words, int policy, void *shared, mcgpu_exchange **out);
int mcgpu_exchange_card(mcgpu_exchange *x, unsigned char *card, size_t card_bytes);
int mcgpu_exchange_connect(mcgpu_exchange *x, int peer, const unsigned char *card, size_t card_bytes);
int mcgpu_exchange_connect_local(mcgpu_exchange *x, mcgpu_exchange *peer);
/* after all peers are connected, before the first step: one small copy-engine transfer into every peer's landing buffer, waited
 * for -- a platform without that path between two devices fails here, where the ranks can still agree on another route */
int mcgpu_exchange_probe(mcgpu_exchange *x);
int mcgpu_exchange_owner(const mcgpu_exchange *x, long long step);
int mcgpu_exchange_begin(mcgpu_exchange *x, long long step, void *hip_stream, void **tally_dev);
int mcgpu_exchange_submit(mcgpu_exchange *x, long long step, void *hip_stream);
int mcgpu_exchange_collect(mcgpu_exchange *x, long long step, void *hip_stream, void **reduced_dev);
/* out6 = {last push [ms], last fused add [ms], pushes, collects, host seconds spent waiting for peers, bytes per push} */
int mcgpu_exchange_stats(mcgpu_exchange *x, double out6[6]);
void mcgpu_exchange_destroy(mcgpu_exchange *x);

/* The vendor-collective route of the same sum (reduce_rccl.cpp): one communicator per device of this process (ncclCommInitAll),
 * then per projection ONE ncclReduce(uint64, sum, root) of the devices' tallies -- the reference's MPI_Reduce(MPI_UNSIGNED_LONG_LONG,
 * MPI_SUM, root 0) of MC-GPU_v1.3.cu:1019 without the trip through the host.  RCCL is opened on first use (dlopen), never linked.
 * create: 0, or -1 when the route cannot be taken here (no library, a device listed twice, no path between the devices) with the
 * reason in mcgpu_last_error().  reduce: tallies[g] is uint64[words] on device g; the sum lands in tallies[root]; the call is
 * enqueued on hip_streams[g] (one thread drives all devices: one group). */
typedef struct mcgpu_rccl mcgpu_rccl;
int mcgpu_rccl_create(const int *devices, int n, mcgpu_rccl **out);
int mcgpu_rccl_reduce_u64(mcgpu_rccl *r, void *const *tallies, size_t words, int root, void *const *hip_streams);
void mcgpu_rccl_destroy(mcgpu_rccl *r);

/* Replace the context's geometry by warp(base geometry, displacement) WITHOUT leaving the device: what
 * MCSimulation4D does per respiratory state with `MCGeometry.warp` + a new voxel file + a new engine process
 * (cbctmc/mc/simulation.py:643-692, cbctmc/mc/geometry.py:386-439).  The base geometry is the one resident at the first
 * call.  displacement: host floats in voxels; frame 0: [3][nz][ny][nx] in the engine's frame (components x, y, z); frame 1:
 * [3][gx][gy][gz] in the frame of the reference's MCGeometry arrays, of which the engine volume is the rot90(k=3) in the x/y
 * plane (cbctmc/mc/geometry.py:589-599) -- the field CorrespondenceModel.predict returns, as it is.  Nearest neighbour with
 * the float32 arithmetic of the grid_sample(nearest, align_corners=True) the reference calls, evaluated in the field's own frame
 * (tests/golden/warp_kat.npz); voxels sampled from outside get (default_material,
 * default_density), which must be in the palette (air at 0.0013 always is).  Palette index volume, both brick levels, the
 * object box and the Woodcock majorant are rebuilt on the device (+ a 24001-entry table on the host); needs a palette
 * volume (<= 256 (material, density) pairs), else -5: fall back to mcgpu_warp_volume + mcgpu_set_geometry_arrays.
 * Dose tallies keep accumulating across such changes. */
int mcgpu_warp_geometry(mcgpu_ctx *ctx, const float *displacement, int frame, int default_material, float default_density);

/* ---- 4-D: one resident context for many (geometry, projection angles) jobs (cbctmc/mc/simulation.py:527-710 launches the
 * engine once per respiratory state) ----
 * mcgpu_set_projection_angles: the explicit-angle list of SECTION ANGLES OF PROJ (MC-GPU_v1.3.cu:1484-1533) replaced at run
 *   time; poses are rebuilt as set_CT_trajectory does (:3280-3434) -- pose 0 stays the input file's (:3313), which is why the
 *   reference passes the first angle twice (simulation.py:658-660); skip it with mcgpu_scan_options::first_projection = 1.
 * mcgpu_set_geometry_arrays: replace the voxel volume from arrays ([z][y][x], as mcgpu_write_voxel_file takes them); densities
 *   pass through the "%.6f" of the voxel file, the material tables are rebuilt (their Woodcock majorant depends on the
 *   volume) and everything is uploaded again.
 * mcgpu_warp_volume: nearest-neighbour warp of (material, density) by a displacement field [3][nz][ny][nx] in voxel units,
 *   out[x] = in[rint(x + u(x))], default outside (geometry.py:386-439: nearest-neighbour grid sampling of the vroc package), on the
 *   context's GPU. */
int mcgpu_set_projection_angles(mcgpu_ctx *ctx, int n, const float *angles_deg);
int mcgpu_set_geometry_arrays(mcgpu_ctx *ctx, const int n[3], const float spacing_cm[3], const uint8_t *material, const float *density);
int mcgpu_warp_volume(mcgpu_ctx *ctx, const int n[3], const uint8_t *material, const float *density, const float *displacement,
                      int default_material, float default_density, uint8_t *material_out, float *density_out);

/* Voxel geometry writer (cbctmc/mc/voxel_data.pyx:12-72 + mcgpu_geometry.jinja2 header fields):
 * material/density are [z][y][x] contiguous, spacing in cm. */
int mcgpu_write_voxel_file(const char *path, const int n[3], const float spacing_cm[3], const uint8_t *material, const float *density,
                           int gzip);

/* Binary sidecar of a voxel file (`geometry.vox[.gz]` -> `geometry.voxbin`): the arrays the text parse would yield (densities
 * quantised through "%.6f" exactly like cbctmc/mc/voxel_data.pyx:25 + MC-GPU_v1.3.cu:2117), palette-compressed.  mcgpu_create
 * prefers a sidecar that is not older than the text file: a 512^3 volume loads in well under a second instead of the
 * 134 M-line text parse per process launch. */
int mcgpu_write_voxel_binary(const char *path, const int n[3], const float spacing_cm[3], const uint8_t *material, const float *density);

/* Hardware ceilings the measurement prices the FAST kernel against, measured on the context's device (about 20 ms each; SURVEY.md
 * 8d; no reference counterpart).  MCGPU_MICROBENCH_VALU_ISSUE: out[0..2] = vector wave-instructions per ns and SIMD of a dense
 * dependent-FMA kernel at 8 waves/SIMD with 64 active lanes, with lanes 0-31, with 32 lanes spread over the wave.
 * MCGPU_MICROBENCH_ATOMIC_RATE: out[0] = scattered 64-bit atomic adds per second into a detector-sized (45 MB) tally. */
#define MCGPU_MICROBENCH_VALU_ISSUE 0
#define MCGPU_MICROBENCH_ATOMIC_RATE 1
int mcgpu_microbench(mcgpu_ctx *ctx, int kind, double *out, int n_out);

/* Device-side known-answer hooks used by the parity tests (each runs a tiny kernel on the context's device). */
int mcgpu_kat_rng(mcgpu_ctx *ctx, int mode, int seed, int batch, int hpt, int n, float *out_f32);
/* Raw 32-bit outputs of the FAST personality's per-history streams (no reference counterpart: the reference's RANECU,
 * MC-GPU_kernel_v1.3.cu:841-894, is the COMPAT personality's).  out_u32[i * n_draws + k] = k-th output of the stream of history
 * ids[i] (ids == NULL: first_id + i) at projection `projection`.  generator 0 = production (Philox4x32-7 seeds a multiply-with-
 * carry lane generator; restated in oracle/fast_rng.py), 1 = Philox4x32-10 per draw (the yardstick of the statistical tests). */
int mcgpu_kat_rng_streams(mcgpu_ctx *ctx, int generator, unsigned int seed, unsigned int projection, unsigned long long first_id,
                          const unsigned long long *ids, int n_ids, int n_draws, uint32_t *out_u32);
int mcgpu_kat_math(mcgpu_ctx *ctx, int n, const double *x, double *out_log, double *out_exp, double *out_sin, double *out_cos);
/* expf as the COMPAT kernel evaluates it (the C library's single-precision algorithm, track_common.inc gl_expf) */
int mcgpu_kat_expf(mcgpu_ctx *ctx, int n, const float *x, float *out_exp);
/* float operations of the COMPAT kernel: op 0 its lean square root, 1 sqrtf, 2 its lean quotient a/b, 3 a/b, 4 shell_pz(a, b, inout) */
int mcgpu_kat_f32(mcgpu_ctx *ctx, int op, int n, const float *a, const float *b, float *inout);
/* Double-precision helpers of MCGPU_MODE_FAST_F64 (csrc/track_fast64.hip), item i -> out8[8 i ..]: sin and cos of
 * 2 pi (u[i] + 1/2) 2^-32; 1 / sqrt(a[i]); sqrt(a[i] / b[i]); cdt1 of MC-GPU_kernel_v1.3.cu:1329 at tau = (float)a[i], E = (float)b[i] 1e5;
 * the direction dir3[3 i ..] rotated by polar cosine c[i] and the azimuth of u[i] (rotate_double, :1103-1148). */
int mcgpu_kat_fast64(mcgpu_ctx *ctx, int n, const uint32_t *u, const double *a, const double *b, const double *c, const float *dir3, double *out8);
/* The 16-byte record of a 4x4x4 tile of a u8-palette volume as the host and the device build it (csrc/device_model.hpp:
 * encode_tile_record; no reference counterpart -- the reference gathers the voxel itself, MC-GPU_kernel_v1.3.cu:262-266).  Host code
 * only, no context: indices[t * 64 + v] = palette index of voxel v = (iz & 3) 16 + (iy & 3) 4 + (ix & 3) of tile t, negative = padding
 * of an edge tile; out_u32[t * 4 ..] = {entries a | b << 8 | c << 16 | d << 24, code, mask low, mask high}. */
int mcgpu_kat_tile_records(int n_tiles, const short *indices, uint32_t *out_u32);

/* ------------------------------------------------------------------------------------------------
 * Row f4: FDK reconstruction of a projection stack (what the reference obtains from `rtkfdk --hardware cuda`,
 * cbctmc/reconstruction/reconstruction.py:22-69).  RTK geometry conventions (rotation axis y, source at
 * Ry(gantry) (0,0,sid), detector coordinate u = sdd x'/(sid - z') - proj_offset_x); projections are line integrals
 * [n_proj][nv][nu] on the host, pixel (i, j) centred at (u0 + i du, v0 + j dv); the volume is written [nz][ny][nx],
 * voxel (0,0,0) centred at (ox, oy, oz) (NaN = volume centred on the isocentre).  hann / hann_y: cut-off of the Hann
 * windows as fractions of Nyquist (0 = plain ramp / no vertical smoothing); wpc: optional water pre-correction
 * polynomial coefficients (rtkfdk --wpc).  Displaced (half-fan) detectors are weighted automatically. */
typedef struct mcgpu_fdk_options {
  unsigned int struct_size;     /* sizeof(mcgpu_fdk_options) as the caller was compiled (later fields read as zero); 0 is refused */
  int n_proj, nu, nv;
  double du, dv, u0, v0;
  double sid, sdd;
  const double *gantry_deg;     /* [n_proj] */
  const double *proj_offset_x;  /* [n_proj] or NULL (0) */
  const double *proj_offset_y;  /* [n_proj] or NULL (0) */
  int nx, ny, nz;
  double sx, sy, sz, ox, oy, oz;
  double hann, hann_y;
  const double *wpc;
  int n_wpc;
  int device;
  double pad;                   /* rtkfdk --pad (truncation correction; the reference passes 1.0, reconstruction.py:29,55): rows
                                   continued on both sides by ceil(pad x width) columns, feathered point reflection; 0 = off.
                                   Follows the PUBLISHED heuristic (Ohnesorge et al.) as RTK describes it; the exact extent and
                                   weight table of rtkFFTProjectionsConvolutionImageFilter could not be checked here (RTK is not
                                   vendored in the reference): with a truncated half-fan scan the feathered edge values fed to the
                                   ramp may differ from rtkfdk's ("parity unpinned") */
} mcgpu_fdk_options;
typedef struct mcgpu_fdk_report {
  double ms_filter;      /* weight + ramp + vertical smoothing kernels */
  double ms_backproject; /* back-projection kernels */
} mcgpu_fdk_report;
int mcgpu_fdk_reconstruct(const mcgpu_fdk_options *options, const float *projections, float *volume, mcgpu_fdk_report *report);

#ifdef __cplusplus
}
#endif
#endif /* MCGPU_AMD_H_ */
