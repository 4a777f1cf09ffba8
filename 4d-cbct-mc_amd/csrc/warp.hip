// warp.hip -- nearest-neighbour warp of a voxelised geometry by a dense displacement field.
//
// The reference warps materials and densities per respiratory state with vroc's SpatialTransformer
// (cbctmc/mc/geometry.py:386-439; vroc is a third-party dependency that is not vendored in the reference tree): identity
// grid + displacement, normalised to [-1, 1] per axis, torch.nn.functional.grid_sample(mode="nearest", align_corners=True),
// voxels sampled from outside the volume get a default (air).  Restated with torch's own float32 steps, because the
// detour through normalised coordinates moves ties and border samples (out[x] = in[rint(x + u)] differs in 212 of the
// 2.1e5 known answers of tests/golden/warp_kat.npz, generated with torch by oracle/gen_warp_golden.py):
//     t = 2 * ((x + u) / (n - 1) - 0.5);   s = ((t + 1) / 2) * (n - 1);   source voxel = nearbyint(s), inside iff 0 <= s' <= n - 1
// every operation rounded to float32 (IEEE division, no contraction: this file is built with -ffp-contract=off).
// HBM-bound gather: 12 B of field + 5 B read + 5 B written per voxel.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mcgpu {
namespace {

__global__ __launch_bounds__(256) void warp_kernel(int nx, int ny, int nz, const unsigned char* __restrict__ mat, const float* __restrict__ dens,
                                                   const float* __restrict__ dvf, unsigned char default_mat, float default_dens,
                                                   unsigned char* __restrict__ out_mat, float* __restrict__ out_dens) {
  const size_t nvox = (size_t)nx * ny * nz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
    auto sample = [](float loc, int n) {
      const float t = 2.0f * (__fdiv_rn(loc, (float)(n - 1)) - 0.5f);
      return rintf(((t + 1.0f) / 2.0f) * (float)(n - 1));
    };
    const float sx = sample((float)x + dvf[i], nx), sy = sample((float)y + dvf[nvox + i], ny), sz = sample((float)z + dvf[2 * nvox + i], nz);
    unsigned char m = default_mat;
    float d = default_dens;
    if (sx >= 0.f && sx <= (float)(nx - 1) && sy >= 0.f && sy <= (float)(ny - 1) && sz >= 0.f && sz <= (float)(nz - 1)) {
      const size_t s = (size_t)(int)sx + (size_t)(int)sy * nx + (size_t)(int)sz * nx * ny;
      m = mat[s];
      d = dens[s];
    }
    out_mat[i] = m;
    out_dens[i] = d;
  }
}

}  // namespace

hipError_t launch_warp(int nx, int ny, int nz, const unsigned char* mat, const float* dens, const float* dvf, unsigned char default_mat,
                       float default_dens, unsigned char* out_mat, float* out_dens, hipStream_t stream) {
  const size_t nvox = (size_t)nx * ny * nz;
  const unsigned blocks = (unsigned)std::min<size_t>((nvox + 255) / 256, 256u * 64u);
  hipLaunchKernelGGL(warp_kernel, dim3(blocks), dim3(256), 0, stream, nx, ny, nz, mat, dens, dvf, default_mat, default_dens, out_mat, out_dens);
  return hipGetLastError();
}

}  // namespace mcgpu
