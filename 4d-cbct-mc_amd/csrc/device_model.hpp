// device_model.hpp -- device-resident tables of the photon-history kernel and its launch arguments.
//
// HBM layout (DESIGN.md "Data layout"):
//   volume    : palette index per voxel, x fastest (u8 when the volume holds <=256 distinct
//               (material,density) pairs, u16 up to 65536, else raw {density, material} float2).
//               512^3 Catphan604 -> 128 MiB (fits the 256 MiB Infinity Cache) instead of the
//               reference's 1 GiB float2 array (MC-GPU_v1.3.cu:2135-2137).
//   palette   : float2 {density, bits(compact material index)}  (staged in LDS when <=256 entries)
//   bricks    : u8 per brick of (2^k)^3 voxels, <= 32768 bricks, LDS-resident: the brick's palette index when all
//               its voxels agree, else 0xFF ("mixed": read the voxel).  Most Woodcock steps land in homogeneous
//               bricks (air, water body) and never touch the volume.
//   mfp       : per (energy bin, compact material) one 32-byte record
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
constexpr int kTrackBlockThreads = 512;   // 8 waves per workgroup
constexpr int kMaxBricks = 32768;         // LDS budget of the brick grid (bytes)

struct TrackArgs {
  // geometry
  const void* vol;
  const float* palette;  // float2 pairs {density, bits(mat_c)}
  int vol_kind, palette_size;
  const unsigned char* bricks;  // brick grid: palette index of a homogeneous brick, 0xFF = mixed (u8 volumes only)
  int brick_shift, brick_nx, brick_nxy, brick_count;
  int nx, ny, nz, nxy;
  float inv_vs[3];
  float bbox[3];
  // energy grid and cross sections
  float e0, ide;
  int num_values, nmat;
  const float* woodcock;  // float2[num_values]
  const float* mfp;       // 8 floats per (bin*nmat + mc)
  // Rayleigh / Compton sampling tables (compact material index)
  const float *xco, *pco, *aco, *bco;
  const unsigned char *itl, *itu;
  const float *fco, *uico, *fj0;  // [shell*nmat + mc]
  const int* noscco;              // [nmat]
  // spectrum
  int nbins;
  const float *espc, *cutoff;
  const short* alias;
  // pose of this projection
  SourcePose src;
  DetectorPose det;
  // tally
  unsigned long long* image;
  // schedule
  int seed, hpt;
  unsigned long long first, count;
  unsigned int stream_key;  // FAST: projection index mixed into the Philox key
  // parked lanes per wave64 that trigger a batched service of that kind
  int thresh_compton, thresh_rayleigh, thresh_new;
  unsigned long long* stats;  // diagnostic build only (8 counters), else null
  unsigned long long* work_counter;  // FAST: next unassigned history offset (zeroed before each launch)
};

}  // namespace mcgpu
