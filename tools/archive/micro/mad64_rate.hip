// One Philox round half as the compiler emits it today (v_mul_hi_u32 + v_mul_lo_u32 + xors) against the same products from ONE
// v_mad_u64_u32 on MI355X: dependent chains of 32 links, 8 waves per SIMD (round 6: is the 64-bit multiply-add worth spelling out in
// rng_init_history?).  hipcc --offload-arch=gfx950 -O3 tools/archive/micro/mad64_rate.hip -o /tmp/mad64_rate && /tmp/mad64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(1024) void chain(unsigned int* out, int iters) {
  unsigned int a = threadIdx.x * 2654435761u + 12345u, b = threadIdx.x | 1u;
  float f = (float)threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (OP == 0) { const unsigned int hi = __umulhi(a, 0xD2511F53u), lo = a * 0xD2511F53u; a = hi ^ b; b = lo; }
      else if (OP == 1) { const unsigned long long p = (unsigned long long)a * 0xD2511F53ULL; unsigned int lo = (unsigned int)p, hi = (unsigned int)(p >> 32); asm volatile("" : "+v"(lo), "+v"(hi)); a = hi ^ b; b = lo; }
      else if (OP == 2) f = fmaf(f, 1.0000001f, 1e-7f);
      else if (OP == 3) { a = a ^ b; b = b ^ (unsigned)k; }
    }
  }
  if (a == 123456u || f == 123.456f || b == 77u) out[0] = a;
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
  printf("mul_hi + mul_lo + xor   %.3f ms\n", run<0>(out));
  printf("mad_u64_u32 + xor       %.3f ms\n", run<1>(out));
  printf("fma_f32                 %.3f ms\n", run<2>(out));
  printf("two xors                %.3f ms\n", run<3>(out));
  return 0;
}
