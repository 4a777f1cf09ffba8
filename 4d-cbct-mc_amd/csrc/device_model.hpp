// device_model.hpp -- device-resident tables of the photon-history kernel and its launch arguments.
//
// HBM layout (DESIGN.md "Data layout"):
//   volume    : palette index per voxel (u8 when the volume holds <=256 distinct (material,density) pairs, u16 up to
//               65536, else raw {density, material} float2).  512^3 Catphan604 -> 128 MiB (fits the 256 MiB Infinity
//               Cache) instead of the reference's 1 GiB float2 array (MC-GPU_v1.3.cu:2135-2137).
//               u8 volumes are stored in TILES of 4x4x4 voxels: tile t = sub-brick index (x fastest over the sub-brick
//               grid), voxel (ix,iy,iz) at byte 64 t + (iz&3) 16 + (iy&3) 4 + (ix&3) (tiled_voxel below).  A tile is one
//               64-byte memory sector and exactly one sub-brick of the second brick level, so the only tiles ever fetched are
//               the MIXED sub-bricks -- the material boundaries -- and each arrives whole: with x-fastest rows a sector was a
//               64x1x1 needle, a mixed sub-brick touched 16 of them, and the fetched set was many times the boundary voxels
//               (thorax: 963 B of fabric traffic per history against 356 algorithmic, round-2 PMC).  u16 / raw: x fastest.
//   palette   : float2 {density, bits(compact material index)}  (staged in LDS when <=256 entries)
//   bricks    : 4 bits per brick of (2^k)^3 voxels, <= 32768 bricks (16 KiB), LDS-resident: code c < 14 when all
//               voxels of the brick share one palette entry (brick_palette[c], the 14 most frequent such entries),
//               14 = EXTERIOR (background brick outside the bounding box of everything else), else 0xF ("mixed":
//               read the voxel).  Most Woodcock steps land in homogeneous bricks (air, water
//               body) and never touch the volume.
//   sub       : the same 4-bit codes per sub-brick of 4^3 voxels, dense over the volume (0.5 MB for 512x512x256), L2-resident:
//               asked by a flight step that lands in a mixed brick before it asks the volume (FAST kernel)
//   mfp_tot   : per (compact material, energy bin) -- material-major rows -- float2 {a_tot, b_tot}: 1.9 MB, L2-resident; the FAST flight step reads
//               only this (virtual-or-real test); the kind of a real interaction is drawn later, in a batch
//   mfp       : per (compact material, energy bin) one 32-byte record
//               {a_tot, a_Co, a_Ra, b_tot | b_Co, b_Ra, pmax(bin+1), 0}  -- one aligned 32-B fetch where the
//               reference reads 2 x float3 from two 7.2 MB tables plus pmax from a third (K.cu:268-269,336).
//   woodcock  : float2 {a,b} per energy bin (K.cu:228)
//   rayleigh  : xco,pco,aco,bco f32[128*nmat], itl,itu u8[128*nmat] (compact materials only)
//   compton   : fco,uico,fj0 f32[40][nmat] (shell-major: lanes in different materials hit different LDS banks)
#pragma once
#include <cstdint>

#include "host_model.hpp"

namespace mcgpu {

enum VolumeKind : int { kVolU8 = 0, kVolU16 = 1, kVolRaw = 2 };
// byte offset of voxel (ix, iy, iz) in a tiled u8 volume whose sub-brick grid is sub_nx wide and sub_nxy per slab
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned int tiled_voxel(unsigned int ix, unsigned int iy, unsigned int iz, unsigned int sub_nx, unsigned int sub_nxy) {
  return ((((ix >> 2) + (iy >> 2) * sub_nx + (iz >> 2) * sub_nxy)) << 6) | ((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u);
}
// Second level of a u8 volume as 16-byte TILE RECORDS (FAST kernel, round 4): the voxels of a 4x4x4 tile take at most two palette
// entries a, b (a material boundary) in all but a few tiles: voxel v (bit (iz&3) 16 + (iy&3) 4 + (ix&3) of `mask`) is b where the
// bit is set, else a.  A flight step into a mixed brick reads the tile's record INSTEAD of the voxel byte: the set of cache lines
// such steps touch is every tile of every mixed brick (thorax: 78 % of the tissue voxels, 22 MB of 64-byte tiles against 4 MB of
// L2 per XCD), and at 16 bytes per tile it is a quarter of that.  Records of the 2x2x2 tiles of an 8^3-voxel cube are contiguous
// (one 128-byte line).
// Round 5: tiles with THREE or FOUR entries -- one in 20 on a CT whose bones are classed voxel by voxel (geo.py:138-166), and each
// of them cost a second, dependent load of the voxel byte -- are held in the same 16 bytes where one entry has the majority:
// a = the most frequent entry, `mask` = the voxels that are NOT a, and `code` says which of b, c (d) the k-th of those voxels
// holds (k = its rank among the set bits of the mask): one bit each for three entries (at most 30 such voxels), two bits each for
// four (at most 15).  code bits 31:30 = 0: one bit per voxel (a two-entry tile is the case code == 0), 1: two bits, 3: none of
// these -- ask the tiled volume.
struct TileRecord {
  unsigned int ab;             // a | b << 8 | c << 16 | d << 24
  unsigned int code;           // see above; 0 for a tile of one or two entries
  unsigned long long mask;
};
constexpr unsigned int kTileAskVolume = 0xC0000000u;
// The record of a tile from its 64 palette indices (v[i] < 0: padding of an edge tile, never addressed).  One definition for the
// host (model_device.cpp) and the device (geometry_device.hip: the records of a warped volume).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline TileRecord encode_tile_record(const short* v) {
  int entry[4] = {-1, -1, -1, -1}, count[4] = {0, 0, 0, 0}, n = 0;
  bool many = false;
  for (int i = 0; i < 64; ++i) {
    if (v[i] < 0) continue;
    int k = 0;
    while (k < n && entry[k] != v[i]) ++k;
    if (k == n) {
      if (n == 4) { many = true; break; }
      entry[n++] = v[i];
    }
    ++count[k];
  }
  TileRecord r{0u, 0u, 0ULL};
  if (n == 0) return r;
  if (!many && n <= 2) {  // the round-4 record: a = the first entry met, b = the other one
    r.ab = (unsigned int)entry[0] | ((unsigned int)(n == 2 ? entry[1] : entry[0]) << 8);
    for (int i = 0; i < 64; ++i)
      if (v[i] >= 0 && v[i] != entry[0]) r.mask |= 1ULL << i;
    return r;
  }
  // a = the most frequent entry (the first met among equals), the others keep their order
  int a = 0, others = 0;
  for (int k = 1; k < n; ++k)
    if (count[k] > count[a]) a = k;
  for (int k = 0; k < n; ++k)
    if (k != a) others += count[k];
  if (many || (n == 3 && others > 30) || (n == 4 && others > 15)) {
    r.ab = (unsigned int)entry[0] | ((unsigned int)entry[1] << 8);
    r.code = kTileAskVolume;
    return r;
  }
  int order[3] = {0, 0, 0}, m = 0;
  for (int k = 0; k < n; ++k)
    if (k != a) order[m++] = k;
  r.ab = (unsigned int)entry[a];
  for (int j = 0; j < m; ++j) r.ab |= (unsigned int)entry[order[j]] << (8 * (j + 1));
  const int bits = n == 3 ? 1 : 2;
  r.code = n == 3 ? 0u : (1u << 30);
  int rank = 0;
  for (int i = 0; i < 64; ++i) {
    if (v[i] < 0 || v[i] == entry[a]) continue;
    int j = 0;
    while (entry[order[j]] != v[i]) ++j;
    r.mask |= 1ULL << i;
    r.code |= (unsigned int)j << (bits * rank);
    ++rank;
  }
  return r;
}
static_assert(sizeof(TileRecord) == 16, "one record = 16 bytes, eight per 128-byte line");
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned int tile_record_index(unsigned int tx, unsigned int ty, unsigned int tz, unsigned int rec_nx, unsigned int rec_nxy) {
  return ((((tx >> 1) + (ty >> 1) * rec_nx + (tz >> 1) * rec_nxy)) << 3) | ((tz & 1u) << 2) | ((ty & 1u) << 1) | (tx & 1u);
}
constexpr int kTrackBlockThreads = 512;   // COMPAT kernel: 8 waves per workgroup
constexpr int kPoolBlockThreads = 1024;   // FAST kernel: 16 waves per workgroup, two workgroups (8 waves/SIMD) per CU share two brick grids
constexpr int kPoolParked = 1;            // FAST kernel: histories a lane parks in LDS slots beside the one in its registers
constexpr int kPoolWavesPerSimd = 2 * kPoolBlockThreads / 64 / 4;  // two workgroups per CU (each gets half of the 160 KB of LDS)
constexpr int kMaxBricks = 32768;         // brick grid budget: 4 bits each -> 16 KiB of LDS
constexpr int kNumCounters = 64, kCounterStride = 32;  // FAST: history-id dispensers (u64 each, 256 B apart)
constexpr int kWoodShift = 6;             // FAST: coarse Woodcock bins of 64 table bins (320 eV): 376 floats of LDS
constexpr int kPoolKinds = 4;             // FAST, workgroup-level pool: kinds of work a parked history can wait for
constexpr int kPoolQueueBytes = 64 + kPoolKinds * kPoolBlockThreads * 2;  // control block + rings (LdsLayout::queues)
constexpr int kSlotWords = 12;            // dwords of a parked history in its lane-private LDS slot (FAST kernel)
constexpr int kS0Bins = 1024;            // COMPAT: energy bins of the S0 bounds (TrackCold::s0_bounds)
constexpr int kNumStats = 32;             // scheduler counters of the diagnostic build
constexpr int kWaveTrace = 16384;         // diagnostic build: {hardware id, first and last clock} of up to this many waves follow the counters
constexpr int kDoseMaterials = 1, kDoseVoxels = 2;  // TrackArgs::dose_flags

// Byte offsets of the kernel's dynamic LDS image (track_common.inc: stage_tables).  Sized for the materials and
// palette entries actually in use so that three 512-thread workgroups fit one CU's 160 KiB.
struct LdsLayout {
  int shells;                // float4 {U, J, f, 0} (COMPAT) / {U, J, alias cut-off, alias shell} (FAST) [sum of shells], material after material
  int nosc;                  // int[2 * nmat]: number of shells, then first shell, per material
  int espc, cutoff, alias;   // float[nbins + 1], float[nbins + 1], short[nbins + 1]
  int pal;                   // float2[16 + palette_size]: brick-code entries, then the palette (u8 volumes)
  int brick;                 // u8[brick_bytes]
  int dose_mat;              // u64[25][2]: per-workgroup material-dose accumulators, flushed at kernel end
  int slots;                 // u32[kSlotWords][kPoolParked * kPoolBlockThreads] (FAST kernel only)
  // FAST kernel only: brackets of the total cross section per (coarse energy bin, material), TrackArgs::sig_shift >= 0
  int sig_mid;               // fp16[ncoarse * nmat]: centre of [min, max] of mfp_tot over the coarse bin
  int sig_w;                 // float[ncoarse]: relative half width that covers every material of the bin
  // FAST kernel only: the Woodcock majorant mean free path per COARSE energy bin of 2^kWoodShift table bins = the smallest value
  // the reference's table (MC-GPU_kernel_v1.3.cu:228) takes anywhere in the bin.  Delta tracking is exact for any majorant, so
  // a history may use the coarse one (a few % more virtual interactions at low energies) and the kernel needs no table fetch from
  // memory when a Compton event or a new photon changes the energy: one dependent round trip fewer per service batch.
  int wood;                  // float[ceil(num_values / 2^kWoodShift)]
  // FAST kernel, workgroup-level pool (track_pool.inc: track_wg_kernel): u32 control block {tail[4], head[4], count[4], panic},
  // then one ring of kPoolBlockThreads u16 slot ids per kind of work {flight, Compton, Rayleigh, tally + source}
  int queues;
  int total;                 // bytes
};

// Tables the kernels touch rarely (staging, Rayleigh sampling): kept behind one pointer so that the launch
// arguments -- which the compiler keeps in scalar registers for the whole kernel -- stay within the SGPR file.
struct TrackCold {
  // Rayleigh / Compton sampling tables (compact material index)
  const float *xco, *pco, *aco, *bco;
  const unsigned char *itl, *itu;
  const float *fco, *uico, *fj0;  // [shell*nmat + mc]
  const int* noscco;              // [nmat]
  const float* shell_cut;         // FAST: Walker alias table of the shell weights f_i per material, [shell*nmat + mc]:
  const unsigned char* shell_alias;  //   cut-off of column `shell` and the shell its upper part maps to
  // spectrum
  const float *espc, *cutoff;
  const short* alias;
  const unsigned char* bricks;  // brick grid, two 4-bit codes per byte (u8 volumes only)
  const unsigned short* sig_mid;  // staging sources of LdsLayout::sig_mid / sig_w (null when the brackets are off)
  const float* sig_w;
  const float* wood_coarse;       // staging source of LdsLayout::wood
  const float* woodcock;          // FAST: the reference's Woodcock table float2[num_values] (entry_face_shell only; TrackArgs::woodcock is the COMPAT kernel's)
  int brick_palette[16];        // palette index of brick code c (c < 15)
  // dose tallies (K.cu:418-443, :1547-1563); buffers live for the whole simulation (all projections accumulate)
  unsigned long long* dose_voxels;     // ulonglong2 {Edep * 100, Edep^2} per ROI voxel, x fastest; null = tally off
  unsigned long long* dose_materials;  // ulonglong2 per material number (25 entries); null = tally off
  int dose_roi[6];                     // 0-based inclusive xmin,xmax,ymin,ymax,zmin,zmax
  int material_of_compact[25];         // material number - 1 of compact material index mc
  int shell_first[25];                 // index of the first Compton shell of compact material mc in the LDS shell table
  // FAST kernel, scheduling points only (kept out of the launch arguments = out of the SGPR file)
  float objbox_lo[3], objbox_hi[3];    // object box [cm]: outside it every brick is kBrickExterior
  // ... and so is every brick that lies wholly outside the elliptic cylinder (axis z) ((x - c0)^2 inv0 + (y - c1)^2 inv1 <= 1)
  // the host fits around everything that is not background: bodies in a CBCT volume are round, and a quarter of their bounding
  // box is corner air.  The object region of exterior_hop / source_entry is box AND cylinder; inv = {0, 0}: no cylinder.
  float ell_c[2], ell_inv[2];
  // parked histories per wave64 that trigger a batched service of that kind; service everything well populated when
  // fewer lanes than `flyable_low` can fly; stop for a scheduling point once `swap_batch` more lanes have parked
  int thresh_compton, thresh_rayleigh, thresh_new, flyable_low, swap_batch;
  // COMPAT kernel: rigorous bounds [lo, hi] of S0 = S(E, theta = pi) (K.cu:1300-1314) per (compact material, energy bin of width
  // 1 / s0_inv_w from s0_emin), float2 [mc * kS0Bins + bin]; decides most Compton angle tests without the pass that computes S0
  const float* s0_bounds;
  float s0_emin, s0_inv_w;
  // FAST: bounds of dx / dy of a sampled source direction in the beam frame, i.e. cot(phi) at the two ends of the azimuthal
  // aperture [phi_low, phi_low + D_phi] (MC-GPU_kernel_v1.3.cu:654-667), computed in double and drawn in by 2e-6 of the range.
  // v_sin_f32 / v_cos_f32 are good to ~1e-6: without the clamp a few 1e-7 of the primary photons leave the aperture by up to
  // 3e-4 detector pixels -- visible only at the half-fan beam edge, which coincides with a pixel boundary (column 1024).
  // (-3e38, 3e38) when the aperture does not lie inside (0, pi) or differs between projections: no clamp.
  float fan_ratio_lo, fan_ratio_hi;
  // FAST kernel: a copy of the LDS layout.  The offsets of the tables only the SERVICES read (Compton shells, spectrum, majorant, dose
  // accumulators) are taken from here -- one scalar load where a batch starts -- instead of from the launch arguments, where each
  // would hold a scalar register for the whole kernel (the flight step's offsets stay launch arguments)
  LdsLayout lds;
  // ... and copies of the launch arguments that only the services read (same reason): the 32-byte cross-section records, the energy
  // grid, the volume's extent.  (Not the dose switches: read from memory they cost the kernel 13 vector registers of scratch.)
  const float* mfp;
  float e0, ide;
  float bbox[3];
  int trade_slots;  // lanes of a wave trade their parking slots: bit 0 before flying, bit 1 before the Compton and tally/source services (MCGPU_SLOT_TRADE)
};

struct TrackArgs {
  // geometry
  const void* vol;
  const float* palette;  // float2 pairs {density, bits(mat_c)}
  int vol_kind, palette_size;
  int brick_shift, brick_nx, brick_nxy, brick_bytes;
  const unsigned char* sub;  // FAST: second level, dense over the 4^3-voxel tiles (null: none): 4-bit codes, or TileRecord[] (sub_kind 2)
  int sub_nx, sub_nxy;
  int sub_kind;              // host-side dispatch only: 0 none, 1 four-bit codes, 2 tile records
  int sched_kind;            // host-side dispatch only: FAST scheduler, 0 = per-wave pools, 1 = workgroup-level pool
  int segment_loop;          // host-side dispatch only: FAST per-wave kernel with the flight segment as an inner loop (tissue volumes)
  int rec_nx, rec_nxy;       // tile records: cubes of 2x2x2 tiles per row / per slab (tile_record_index)
  int nx, ny, nz, nxy;
  float inv_vs[3];
  float bbox[3];
  // FAST: region outside the object box (bricks coded kBrickExterior) is homogeneous background (palette slot 14);
  // bit 0: hop during flight, bit 1: hop at the source (the box itself is in TrackCold)
  int has_exterior;
  float bbox_hi[3];  // FAST: largest coordinate still inside: <= bbox - EPS and mapping into the last voxel / brick
  LdsLayout lds;
  // energy grid and cross sections
  float e0, ide;
  int num_values, nmat;
  const float* woodcock;  // float2[num_values]
  const float* mfp;       // 8 floats per row mc*num_values + bin (track_common.inc: table_row)
  const float* mfp_tot;   // float2 {a_tot, b_tot} per row mc*num_values + bin: the only cross section a flight step needs (FAST)
  int sig_shift;          // FAST: coarse energy bin = bin >> sig_shift for the LDS brackets of mfp_tot; -1 = no brackets
  const TrackCold* cold;
  int nbins;
  // pose of this projection (device-resident arrays of all projections, uploaded once)
  const SourcePose* src;
  const DetectorPose* det;
  // tally
  unsigned long long* image;
  // schedule
  int seed, hpt;
  unsigned long long first, count;
  unsigned int stream_key;  // FAST: projection index mixed into the Philox key
  // COMPAT kernel: parked lanes per wave64 that trigger a batched service of that kind
  int thresh_compton, thresh_rayleigh, thresh_new;
  int thresh_take;  // COMPAT: lanes whose parked history could fly while their register history cannot, to exchange the two
  int dose_flags;             // bit 0: material dose tally, bit 1: voxel dose tally (TrackCold holds the buffers)
  unsigned long long* stats;  // diagnostic build only (kNumStats counters), else null
  unsigned long long* work_counter;  // FAST: kNumCounters id dispensers, kCounterStride words apart (zeroed before each launch)
};

}  // namespace mcgpu
