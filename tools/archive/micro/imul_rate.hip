// Issue cost of the 32-bit integer multiplies against an add, an xor-rotate and an FMA on MI355X (gfx950): dependent
// chains of 32 operations, 8 waves per SIMD (the Philox4x32 seeding of the FAST kernel is 28 of those multiplies per history)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(1024) void chain(unsigned int* out, int iters) {
  unsigned int a = threadIdx.x * 2654435761u + 12345u, b = threadIdx.x | 1u;
  float f = (float)threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (OP == 0) a = a + b + (unsigned)k;                       // v_add3 / v_add
      else if (OP == 1) a = a * 0xD2511F53u + b;                  // v_mul_lo_u32 (+ add)
      else if (OP == 2) a = __umulhi(a, 0xD2511F53u) ^ b;         // v_mul_hi_u32 (+ xor)
      else if (OP == 3) a = ((a << 11) | (a >> 21)) ^ b;          // v_alignbit + xor
      else if (OP == 4) f = fmaf(f, 1.0000001f, 1e-7f);           // v_fma_f32
      else if (OP == 5) a = __umul24(a, 0x51F53u) + b;            // v_mad_u32_u24
    }
  }
  if (a == 123456u || f == 123.456f) out[0] = a;
}
template <int OP>
float run(unsigned int* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chain<OP><<<512, 1024>>>(out, 2000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  chain<OP><<<512, 1024>>>(out, 20000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  unsigned int* out; hipMalloc(&out, 4);
  printf("add        %.3f ms\n", run<0>(out));
  printf("mul_lo+add %.3f ms\n", run<1>(out));
  printf("mul_hi^xor %.3f ms\n", run<2>(out));
  printf("rot^xor    %.3f ms\n", run<3>(out));
  printf("fma_f32    %.3f ms\n", run<4>(out));
  printf("mad_u24    %.3f ms\n", run<5>(out));
  return 0;
}
