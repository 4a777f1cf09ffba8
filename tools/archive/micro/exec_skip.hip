// Does a VALU instruction of a wave64 cost less when whole 16-lane groups are inactive?  (MI355X, gfx950)
// A dependent FMA chain runs under different EXEC masks; 8 waves per SIMD keep the VALU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void chain(float* out, int mode, int iters) {
  const int lane = threadIdx.x & 63;
  bool active;
  switch (mode) {
    case 0: active = true; break;               // 64 lanes
    case 1: active = lane < 32; break;          // lower half
    case 2: active = lane < 16; break;          // one 16-lane group
    case 3: active = (lane & 3) == 0; break;    // 16 lanes spread over all groups
    case 4: active = lane < 48; break;
    default: active = (lane & 1) == 0; break;   // 32 lanes spread
  }
  float a = (float)threadIdx.x * 1e-3f, b = 1.0000001f, c = 1e-7f;
  if (active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 32; ++k) a = fmaf(a, b, c);
    }
  }
  if (a == 123.456f) out[0] = a;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"64 lanes", "lanes 0-31", "lanes 0-15", "16 lanes spread", "lanes 0-47", "32 lanes spread"};
  for (int mode = 0; mode < 6; ++mode) {
    chain<<<512, 1024>>>(out, mode, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<<<512, 1024>>>(out, mode, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %.3f ms\n", names[mode], ms);
  }
  return 0;
}
