// geometry_device.hpp -- interface of the on-device geometry rebuild (geometry_device.hip), used by model_device.cpp and engine_geometry.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "device_model.hpp"

namespace mcgpu {

struct GeometryRebuild {
  int nx, ny, nz;
  int brick_shift, bn[3];           // first level: bricks of (2^brick_shift)^3 voxels
  int sn[3];                        // second level: sub-bricks of 4^3 voxels
  const unsigned char* base_idx;    // palette index volume of the base geometry, tiled 4x4x4 like the live one (warp source)
  const float* dvf;                 // displacement field in voxels, 3 x nvox floats (layout: warp_frame)
  unsigned char default_index;      // palette index of the default (material, density) for samples from outside
  unsigned char* idx;               // tiled palette index volume the kernels read from now on (warp destination)
  unsigned short* sub_first;        // scratch [sub-bricks]: palette entry or 0x100 = mixed
  unsigned short* brick_first;      // scratch [bricks]
  unsigned char* sub;               // 4-bit codes of the sub-bricks, two per byte (null: level not in use)
  TileRecord* rec;                  // 16-byte records of the tiles (device_model.hpp; null: not in use)
  int rn[3];                        // cubes of 2x2x2 tiles per axis (tile_record_index)
  unsigned char* bricks;            // 4-bit codes of the bricks, two per byte
  const unsigned char* code_of;     // [256] palette entry -> 4-bit code (0xF: not among the 14 coded entries)
  int background;                   // palette entry of the homogeneous background (outside the object box)
  // 17 words for the host: [0..7] bit v set = palette entry v occurs; [8..10] / [11..13] object box in bricks (lo / hi);
  // [14] mixed bricks, [15] exterior bricks, [16] mixed sub-bricks
  unsigned int* out;
};

// warp (warp_frame 0: field in the engine's frame [3][nz][ny][nx]; 1: in the reference's MCGeometry frame [3][gx][gy][gz];
// < 0: no warp) + classification + code tables, all on `stream`; the caller synchronises and reads `out`
hipError_t launch_geometry_rebuild(const GeometryRebuild& g, int warp_frame, bool allow_exterior, hipStream_t stream);

}  // namespace mcgpu
