// geometry_device.hip -- geometry change of a resident context entirely on the device ("keep tables
// on device, swap/warp volumes on GPU").
//
// The reference handles every respiratory state of a 4-D scan as a new simulation: warp materials and densities on the
// CPU, write a .vox.gz, start the engine, parse 14 M text lines (cbctmc/mc/simulation.py:643-692).  Round 1 of this engine
// warped on the GPU but rebuilt palette, brick grid, object box and Woodcock majorant on the host (150 ms per state).  A
// nearest-neighbour warp only moves voxels around: the set of (material, density) pairs is the base geometry's (plus the
// default, air), so the palette stays and what is warped is the 1-byte palette index volume.  Everything derived from the
// voxels is then recomputed here:
//   warp_index_kernel      out[x] = base[nearest(x + u(x))] or the default's palette index   (torch's arithmetic, warp.hip)
//   classify_sub_kernel    per sub-brick of 4^3 voxels: its palette entry or "mixed"; which palette entries occur at all
//   classify_brick_kernel  per brick of (2^k)^3 voxels: the same from its sub-bricks; bounding box of the non-background bricks
//   pack_codes_kernel      the 4-bit code tables of both levels (EXTERIOR outside the bounding box)
// The host gets back 17 words: the occupancy bits (-> largest density per material -> Woodcock table, 24001 x nmat
// operations), the object box and three counters.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_model.hpp"
#include "geometry_device.hpp"

namespace mcgpu {
namespace {

constexpr unsigned short kMixed = 0x100;

// identity + displacement -> normalised -> grid_sample(align_corners=True) un-normalisation -> nearbyint, in float32 like
// torch (see warp.hip; known answers tests/golden/warp_kat.npz)
__device__ __forceinline__ float nearest_sample(float loc, int n) {
  const float t = 2.0f * (__fdiv_rn(loc, (float)(n - 1)) - 0.5f);
  return rintf(((t + 1.0f) / 2.0f) * (float)(n - 1));
}

// FRAME 0: the field is given in the engine's frame, [3][nz][ny][nx], components (x, y, z).
// FRAME 1: the field is given in the frame of the reference's MCGeometry arrays, [3][gx][gy][gz] with the engine volume =
//          rot90(k=3) of them in the x/y plane (create_mcgpu_geometry, cbctmc/mc/geometry.py:589-599): engine voxel
//          (x, y, z) is geometry voxel (gx, gy, gz) = (ny - 1 - y, x, z).  The warp is evaluated in the geometry's frame,
//          where the reference evaluates it (ties and border samples do not survive a mirrored axis), and only the
//          result is addressed in the engine's layout.
// Source and destination are TILED index volumes (device_model.hpp: tiled_voxel).  One thread per voxel of the padded
// grid, in tile order: a wave writes one whole 64-byte tile (the padding voxels of edge tiles get the default).
template <int FRAME>
__global__ __launch_bounds__(256) void warp_index_kernel(int nx, int ny, int nz, int snx, int sny, int snz, const unsigned char* __restrict__ base,
                                                         const float* __restrict__ dvf, unsigned char default_index, unsigned char* __restrict__ out) {
  const size_t nvox = (size_t)nx * ny * nz, ncell = (size_t)snx * sny * snz * 64;
  const unsigned int snxy = (unsigned int)(snx * sny);
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (size_t)gridDim.x * blockDim.x) {
    const size_t t = c >> 6;
    const int x = ((int)(t % snx) << 2) | (int)(c & 3), y = ((int)((t / snx) % sny) << 2) | (int)((c >> 2) & 3), z = ((int)(t / snxy) << 2) | (int)((c >> 4) & 3);
    unsigned char v = default_index;
    if (x < nx && y < ny && z < nz) {
      if (FRAME == 0) {
        const size_t i = (size_t)x + (size_t)y * nx + (size_t)z * nx * ny;
        const float sx = nearest_sample((float)x + dvf[i], nx), sy = nearest_sample((float)y + dvf[nvox + i], ny), sz = nearest_sample((float)z + dvf[2 * nvox + i], nz);
        if (sx >= 0.f && sx <= (float)(nx - 1) && sy >= 0.f && sy <= (float)(ny - 1) && sz >= 0.f && sz <= (float)(nz - 1))
          v = base[tiled_voxel((unsigned int)(int)sx, (unsigned int)(int)sy, (unsigned int)(int)sz, (unsigned int)snx, snxy)];
      } else {
        const int g0 = ny, g1 = nx, g2 = nz;  // extents of the geometry arrays
        const int gx = ny - 1 - y, gy = x, gz = z;
        const size_t f = ((size_t)gx * g1 + gy) * g2 + gz;
        const float sx = nearest_sample((float)gx + dvf[f], g0), sy = nearest_sample((float)gy + dvf[nvox + f], g1), sz = nearest_sample((float)gz + dvf[2 * nvox + f], g2);
        if (sx >= 0.f && sx <= (float)(g0 - 1) && sy >= 0.f && sy <= (float)(g1 - 1) && sz >= 0.f && sz <= (float)(g2 - 1))
          v = base[tiled_voxel((unsigned int)(int)sy, (unsigned int)(ny - 1 - (int)sx), (unsigned int)(int)sz, (unsigned int)snx, snxy)];
      }
    }
    out[c] = v;
  }
}

// one thread per sub-brick of 4^3 voxels
__global__ __launch_bounds__(256) void classify_sub_kernel(GeometryRebuild g) {
  __shared__ unsigned int seen[8];
  if (threadIdx.x < 8) seen[threadIdx.x] = 0u;
  __syncthreads();
  const int nsub = g.sn[0] * g.sn[1] * g.sn[2];
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nsub) {
    const int sx = s % g.sn[0], sy = (s / g.sn[0]) % g.sn[1], sz = s / (g.sn[0] * g.sn[1]);
    const int x0 = sx << 2, y0 = sy << 2, z0 = sz << 2;
    const int x1 = min(x0 + 4, g.nx), y1 = min(y0 + 4, g.ny), z1 = min(z0 + 4, g.nz);
    int first = -1, second = -1;
    bool mixed = false, many = false;
    unsigned long long mask = 0ULL;  // tile record: voxels that hold the tile's second palette entry
    const unsigned char* tile = g.idx + ((size_t)s << 6);  // the sub-brick's own 64-byte tile
    for (int z = z0; z < z1; ++z)
      for (int y = y0; y < y1; ++y) {
        const unsigned char* row = tile + ((z & 3) << 4) + ((y & 3) << 2) - x0;
        for (int x = x0; x < x1; ++x) {
          const int v = row[x];
          if (first < 0) first = v;
          else if (v != first) {
            mixed = true;
            if (second < 0) second = v;
            if (v == second) mask |= 1ULL << (((z & 3) << 4) | ((y & 3) << 2) | (x & 3));
            else many = true;
          }
          atomicOr(&seen[v >> 5], 1u << (v & 31));
        }
      }
    g.sub_first[s] = mixed ? kMixed : (unsigned short)first;
    if (g.rec) {
      TileRecord r;
      if (!many) {  // one or two entries: the record is already known (what encode_tile_record gives for such a tile)
        r.ab = (unsigned int)(first < 0 ? 0 : first) | ((unsigned int)(second < 0 ? (first < 0 ? 0 : first) : second) << 8);
        r.code = 0u;
        r.mask = mask;
      } else {
        short v64[64];
        for (int i = 0; i < 64; ++i) {
          const int x = x0 + (i & 3), y = y0 + ((i >> 2) & 3), z = z0 + (i >> 4);
          v64[i] = (x >= g.nx || y >= g.ny || z >= g.nz) ? (short)-1 : (short)tile[i];
        }
        r = encode_tile_record(v64);
      }
      g.rec[tile_record_index((unsigned int)sx, (unsigned int)sy, (unsigned int)sz, (unsigned int)g.rn[0], (unsigned int)(g.rn[0] * g.rn[1]))] = r;
    }
  }
  __syncthreads();
  if (threadIdx.x < 8 && seen[threadIdx.x] != 0u) atomicOr(&g.out[threadIdx.x], seen[threadIdx.x]);
}

// one thread per brick of (2^k)^3 voxels, k >= 2: a brick is a whole number of sub-bricks
__global__ __launch_bounds__(256) void classify_brick_kernel(GeometryRebuild g) {
  const int nb = g.bn[0] * g.bn[1] * g.bn[2];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const int bx = b % g.bn[0], by = (b / g.bn[0]) % g.bn[1], bz = b / (g.bn[0] * g.bn[1]);
  const int per = 1 << (g.brick_shift - 2);
  const int sx0 = bx * per, sy0 = by * per, sz0 = bz * per;
  const int sx1 = min(sx0 + per, g.sn[0]), sy1 = min(sy0 + per, g.sn[1]), sz1 = min(sz0 + per, g.sn[2]);
  int first = -1;
  bool mixed = false;
  for (int z = sz0; z < sz1; ++z)
    for (int y = sy0; y < sy1; ++y)
      for (int x = sx0; x < sx1; ++x) {
        const int v = g.sub_first[((size_t)z * g.sn[1] + y) * g.sn[0] + x];
        if (v == kMixed) mixed = true;
        else if (first < 0) first = v;
        else if (v != first) mixed = true;
      }
  g.brick_first[b] = mixed ? kMixed : (unsigned short)first;
  if (mixed || first != g.background) {  // part of the object: grows the object box (in bricks)
    atomicMin((int*)&g.out[8], bx); atomicMin((int*)&g.out[9], by); atomicMin((int*)&g.out[10], bz);
    atomicMax((int*)&g.out[11], bx); atomicMax((int*)&g.out[12], by); atomicMax((int*)&g.out[13], bz);
  }
}

// one thread per byte of a 4-bit code table (two entries); level 0 = bricks, level 1 = sub-bricks
__global__ __launch_bounds__(256) void pack_codes_kernel(GeometryRebuild g, int level, int allow_exterior) {
  const int n = level == 0 ? g.bn[0] * g.bn[1] * g.bn[2] : g.sn[0] * g.sn[1] * g.sn[2];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * t >= n) return;
  const int lo[3] = {(int)g.out[8], (int)g.out[9], (int)g.out[10]}, hi[3] = {(int)g.out[11], (int)g.out[12], (int)g.out[13]};
  const bool have_box = allow_exterior && hi[0] >= lo[0];
  int byte = 0xFF, n_mixed = 0, n_ext = 0;
  for (int h = 0; h < 2; ++h) {
    const int e = 2 * t + h;
    if (e >= n) break;
    int code;
    if (level == 0) {
      const int bx = e % g.bn[0], by = (e / g.bn[0]) % g.bn[1], bz = e / (g.bn[0] * g.bn[1]);
      const bool outside = bx < lo[0] || bx > hi[0] || by < lo[1] || by > hi[1] || bz < lo[2] || bz > hi[2];
      const int f = g.brick_first[e];
      if (have_box && outside) { code = 14; ++n_ext; }
      else code = (f == kMixed) ? 0xF : g.code_of[f];
    } else {
      const int f = g.sub_first[e];
      code = (f == kMixed) ? 0xF : g.code_of[f];
    }
    n_mixed += (code == 0xF);
    byte = (byte & ~(0xF << (4 * h))) | (code << (4 * h));
  }
  (level == 0 ? g.bricks : g.sub)[t] = (unsigned char)byte;
  if (n_mixed) atomicAdd(&g.out[level == 0 ? 14 : 16], (unsigned int)n_mixed);
  if (n_ext) atomicAdd(&g.out[15], (unsigned int)n_ext);
}

__global__ void init_out_kernel(unsigned int* out) {
  const int i = threadIdx.x;
  if (i < 17) out[i] = (i >= 8 && i <= 10) ? 0x7FFFFFFFu : ((i >= 11 && i <= 13) ? 0x80000000u : 0u);
}

}  // namespace

hipError_t launch_geometry_rebuild(const GeometryRebuild& g, int warp_frame, bool allow_exterior, hipStream_t stream) {
  const size_t ncell = (size_t)g.sn[0] * g.sn[1] * g.sn[2] * 64;
  const unsigned wblocks = (unsigned)std::min<size_t>((ncell + 255) / 256, 256u * 64u);
  if (warp_frame == 0)
    hipLaunchKernelGGL(warp_index_kernel<0>, dim3(wblocks), dim3(256), 0, stream, g.nx, g.ny, g.nz, g.sn[0], g.sn[1], g.sn[2], g.base_idx, g.dvf, g.default_index, g.idx);
  else if (warp_frame == 1)
    hipLaunchKernelGGL(warp_index_kernel<1>, dim3(wblocks), dim3(256), 0, stream, g.nx, g.ny, g.nz, g.sn[0], g.sn[1], g.sn[2], g.base_idx, g.dvf, g.default_index, g.idx);
  hipLaunchKernelGGL(init_out_kernel, dim3(1), dim3(32), 0, stream, g.out);
  const int nsub = g.sn[0] * g.sn[1] * g.sn[2], nb = g.bn[0] * g.bn[1] * g.bn[2];
  hipLaunchKernelGGL(classify_sub_kernel, dim3((nsub + 255) / 256), dim3(256), 0, stream, g);
  hipLaunchKernelGGL(classify_brick_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, g);
  hipLaunchKernelGGL(pack_codes_kernel, dim3(((nb + 1) / 2 + 255) / 256), dim3(256), 0, stream, g, 0, allow_exterior ? 1 : 0);
  if (g.sub) hipLaunchKernelGGL(pack_codes_kernel, dim3(((nsub + 1) / 2 + 255) / 256), dim3(256), 0, stream, g, 1, 0);
  return hipGetLastError();
}

}  // namespace mcgpu
