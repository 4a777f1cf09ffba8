// finalize.hip -- device side of the projection post-processing (postprocess.cpp): integer tallies u64[4][Nz][Nx] ->
// float32 planes [3][Nz][crop_nx] {total, unscattered, scattered}, z flipped and half-fan cropped
// (cbctmc/mc/projection.py:42-51,118-127), bit-identical to finalize_projection_host.  Optionally zeroes the tallies
// in the same pass (init_image_array_GPU, MC-GPU_kernel_v1.3.cu:56-72), so a scan needs no separate clear.
// HBM-bound streaming: 32 B read (+32 B written when clearing) and 12 B written per pixel.
// Built with -ffp-contract=off: the arithmetic is plain IEEE double, identical to the host's.
#include <hip/hip_runtime.h>

#include "decimal8.hpp"

namespace mcgpu {
namespace {

__global__ __launch_bounds__(256) void finalize_kernel(unsigned long long* image, int nx, int nz, int crop_nx, double norm, float* planes,
                                                       int clear) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int z = blockIdx.y;  // input row
  if (x >= nx) return;
  const size_t npix = (size_t)nx * nz, src = (size_t)z * nx + x;
  const unsigned long long t0 = image[src], t1 = image[src + npix], t2 = image[src + 2 * npix], t3 = image[src + 3 * npix];
  if (clear) { image[src] = 0ULL; image[src + npix] = 0ULL; image[src + 2 * npix] = 0ULL; image[src + 3 * npix] = 0ULL; }
  if (x < crop_nx) {
    const size_t plane = (size_t)crop_nx * nz, dst = (size_t)(nz - 1 - z) * crop_nx + x;
    float tot, uns, sca;
    finalize_pixel(t0, t1, t2, t3, norm, tot, uns, sca);
    planes[dst] = tot;
    planes[plane + dst] = uns;
    planes[2 * plane + dst] = sca;
  }
}

// dst += src over the tally words (the integer sum of the reference's MPI_Reduce, MC-GPU_v1.3.cu:1019)
__global__ __launch_bounds__(256) void accumulate_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// dst += src[0] + ... + src[n-1] in ONE pass (tally exchange, exchange.cpp): every word of every source is read once, the
// destination is read and written once -- (n + 2) x 45 MB instead of the 3n x 45 MB of n separate adds.  16-byte accesses.
__global__ __launch_bounds__(256) void accumulate_many_kernel(unsigned long long* __restrict__ dst, const unsigned long long* const* __restrict__ src, int n, size_t pairs) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 acc = reinterpret_cast<ulonglong2*>(dst)[i];
    for (int k = 0; k < n; ++k) {
      const ulonglong2 v = reinterpret_cast<const ulonglong2*>(src[k])[i];
      acc.x += v.x; acc.y += v.y;
    }
    reinterpret_cast<ulonglong2*>(dst)[i] = acc;
  }
}

}  // namespace

hipError_t launch_accumulate_many(unsigned long long* dst, const unsigned long long* const* src, int n, size_t words, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (words & 1) return hipErrorInvalidValue;  // tallies are 4 planes of pixels: always even
  hipLaunchKernelGGL(accumulate_many_kernel, dim3(4096), dim3(256), 0, stream, dst, src, n, words / 2);
  return hipGetLastError();
}

hipError_t launch_accumulate(unsigned long long* dst, const unsigned long long* src, size_t words, hipStream_t stream) {
  hipLaunchKernelGGL(accumulate_kernel, dim3(2048), dim3(256), 0, stream, dst, src, words);
  return hipGetLastError();
}

hipError_t launch_finalize(unsigned long long* image, int nx, int nz, int crop_nx, double norm, float* planes, int clear, hipStream_t stream) {
  const dim3 block(256), grid((unsigned)((nx + 255) / 256), (unsigned)nz);
  hipLaunchKernelGGL(finalize_kernel, grid, block, 0, stream, image, nx, nz, crop_nx, norm, planes, clear);
  return hipGetLastError();
}

}  // namespace mcgpu
