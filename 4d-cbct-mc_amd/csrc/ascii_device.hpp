// ascii_device.hpp -- interface of the on-device formatter of the projection files' data lines (ascii_device.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace mcgpu {

struct AsciiArgs {
  const unsigned long long* image;  // uint64[4][nz][nx] tallies
  int nx, nz;
  size_t npix;
  double norm;                      // report_image's NORM (MC-GPU_v1.3.cu:2860-2861)
  char* text;                       // out: the data lines, rows separated by a blank line
  unsigned long long capacity;      // bytes available at `text`
  // per detector row (nz entries; row_off has nz + 1: the last one is the total length)
  unsigned long long *row_len, *row_off;
  double *row_sum, *row_max;        // sum of the four classes over the row / its largest pixel sum ...
  long long* row_arg;               // ... and that pixel's index (first of equal maxima)
  unsigned int* flags;              // bit 0: a value outside the formatter's range, bit 1: text buffer too small
};

hipError_t launch_ascii_format(const AsciiArgs& a, hipStream_t stream);

}  // namespace mcgpu
