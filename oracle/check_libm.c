/* TEST INFRASTRUCTURE.  Compares the oracle's portable expf (gl_expf) with the host C library's expf() on EVERY float:
 *   gcc -O2 -fopenmp check_libm.c -L. -loracle -Wl,-rpath,'$ORIGIN' -lm -o /tmp/check_libm && /tmp/check_libm
 * (about 20 s on 8 cores).  Result on the build machine (glibc 2.35, x86-64 with FMA): 0 of 4 278 190 082 differ. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
float oracle_gl_expf(float x);
int main(void)
{
  unsigned long long n = 0, bad = 0;
#pragma omp parallel for reduction(+ : n, bad)
  for (long long u = 0; u < 0x100000000LL; ++u) {
    uint32_t w = (uint32_t)u;
    float x, a, b;
    memcpy(&x, &w, 4);
    if (x != x) continue;
    ++n;
    a = expf(x);
    b = oracle_gl_expf(x);
    if (memcmp(&a, &b, 4) != 0) ++bad;
  }
  printf("expf: %llu floats, %llu differ from the host libm\n", n, bad);
  return bad != 0;
}
