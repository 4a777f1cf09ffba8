/*
 * oracle/mcgpu_oracle.c -- TEST INFRASTRUCTURE ONLY (see mcgpu_oracle.h).
 *
 * CPU restatement of the reference photon-history loop.  Every function cites the lines of
 * docker/mcgpu/MC-GPU_kernel_v1.3.cu ("K.cu") it follows; the branches taken are those of the
 * reference's CPU build (USING_CUDA undefined).  Operand types (float vs double) and operation
 * order are kept as the C expressions of the reference evaluate them, so that with
 * ORACLE_MATH_LIBM the tallies are bit-identical to oracle/_ref.  Compile with
 * -ffp-contract=off and without fast-math (oracle/Makefile).
 *
 * ORACLE_MATH_PORTABLE swaps the libm calls inside the history loop (logf, expf, powf, sin,
 * cos) for the deterministic functions below ("pm_": IEEE +,-,*,/ on doubles and bit moves;
 * "gl_expf": the C library's own single-precision algorithm with explicit fma); the HIP compat
 * kernel restates the same functions, which makes GPU-vs-oracle comparisons bit-exact.
 */
#include "mcgpu_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXMAT 25
#define MAXSHELLS 40
#define NPRAY 128
#define EPS_SRC 0.000015f
#define NEG_INF_DIST (-500000.0f)

typedef struct { float x, y, z; } f3;
typedef struct { int x, y; } i2;

/* reference wire layouts (MC-GPU_v1.3.h:155-169 and :190-208) */
typedef struct {
  f3 position, direction;
  float rot_fan[9], cos_theta_low, phi_low, D_cos_theta, D_phi, max_height_at_y1cm;
} src_t;
typedef struct {
  float sdd, lateral_displacement;
  f3 corner_min_rotated_to_Y, center;
  float rot_inv[9], width_X, height_Z, inv_pixel_size_X, inv_pixel_size_Z;
  int num_pixels_x, num_pixels_y, total_num_pixels, rotation_flag;
} det_t;

/* ------------------------------------------------------------------------------------------
 * Portable math.  Algorithms: log via a 64-entry table and the Taylor polynomial of log1p (pm_log below);
 * exp via Cody-Waite reduction by ln2 and a degree-13 Taylor polynomial; sin/cos via reduction
 * by pi/2 (two-part constant) and Taylor polynomials on |r| <= pi/4.  No fma, no libm.  (pm_exp now serves powf only:
 * expf of this mode is gl_expf below.)
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ln of a positive normal double: x = 2^k z with z in [OFF, 2 OFF), OFF ~ sqrt(1/2); a 64-entry table {1/c_i, -ln(1/c_i)} for
 * the interval of z (oracle/gen_log_table.py: pure arithmetic; c = 1 for the interval around 1.0), r = z/c - 1 by one fma,
 * |r| < 0.008, log1p(r) by its Taylor polynomial to r^8 (error r^9/9 < 2e-20).  No division (round 2's 2 atanh series had
 * one: a third of the instructions of a Woodcock step on the device). */
static const double PM_LOG_T[64][2] = {
  {0x1.680cbcd75c447p+0, -0x1.5d401699901fbp-2}, {0x1.6422f70fce190p+0, -0x1.520f66629c64ap-2},
  {0x1.604ebbd86ed05p+0, -0x1.46fdadb7c4a33p-2}, {0x1.5c8f5b36b5f7ap+0, -0x1.3c0a43098441cp-2},
  {0x1.58e42c98c7254p+0, -0x1.3134822a443dfp-2}, {0x1.554c8e72ab8d1p+0, -0x1.267bcc14a5204p-2},
  {0x1.51c7e5e1a7890p+0, -0x1.1bdf86b4c6653p-2}, {0x1.4e559e553c622p+0, -0x1.115f1cb45c41bp-2},
  {0x1.4af5293d6fae2p+0, -0x1.06f9fd496801ep-2}, {0x1.47a5fdbdf9ca4p+0, -0x1.f95f380ed4a55p-3},
  {0x1.44679866047a0p+0, -0x1.e4fee165d342bp-3}, {0x1.41397aec297ffp+0, -0x1.d0d1ee2e52a3ep-3},
  {0x1.3e1b2bee6741cp+0, -0x1.bcd75dbbe1f04p-3}, {0x1.3b0c36b5c7393p+0, -0x1.a90e36d289f47p-3},
  {0x1.380c2afd77253p+0, -0x1.9575875de8192p-3}, {0x1.351a9cbd1ab0bp+0, -0x1.820c642bbdb66p-3},
  {0x1.323723f61fa13p+0, -0x1.6ed1e8a9c1f43p-3}, {0x1.2f615c83e29d7p+0, -0x1.5bc536a687a6cp-3},
  {0x1.2c98e5ee7644dp+0, -0x1.48e576154b88ep-3}, {0x1.29dd633fe1b19p+0, -0x1.3631d4d4821b2p-3},
  {0x1.272e7adbae92dp+0, -0x1.23a98676fee40p-3}, {0x1.248bd658a1e75p+0, -0x1.114bc40f914ebp-3},
  {0x1.21f5225c7cfd0p+0, -0x1.fe2f97fdeb17dp-4}, {0x1.1f6a0e79a6c4ep+0, -0x1.da19c387f3f66p-4},
  {0x1.1cea4d0e9fc17p+0, -0x1.b6549b9b77017p-4}, {0x1.1a75932724e42p+0, -0x1.92deba9fba046p-4},
  {0x1.180b985ee791fp+0, -0x1.6fb6c4313fd6bp-4}, {0x1.15ac16c5c2c72p+0, -0x1.4cdb64d18a57ep-4},
  {0x1.1356cac556ee6p+0, -0x1.2a4b519a3f19bp-4}, {0x1.110b7307f780bp+0, -0x1.080547f38348dp-4},
  {0x1.0ec9d060d6e24p+0, -0x1.cc101a9acb805p-5}, {0x1.0c91a5b55e3b4p+0, -0x1.88a4ddb8616c5p-5},
  {0x1.0a62b7e7a03aap+0, -0x1.45c682aefd356p-5}, {0x1.083ccdc1d6c93p+0, -0x1.0372c177df6e4p-5},
  {0x1.061fafe2dcbc4p+0, -0x1.834ec03d80f5dp-6}, {0x1.040b28ab9586fp+0, -0x1.00c4649e005fbp-6},
  {0x1.01ff042d35cc6p+0, -0x1.fd08c72af8ec1p-8}, {0x1.0000000000000p+0, 0x0.0p+0},
  {0x1.f80c5d11a2684p-1, 0x1.0073809b61d99p-6}, {0x1.f0698d819db27p-1, 0x1.fa8e7ea7db662p-6},
  {0x1.e901160cae199p-1, 0x1.7873639e2b330p-5}, {0x1.e1d063d89c40fp-1, 0x1.f1cc700489fe1p-5},
  {0x1.dad50a3b936f0p-1, 0x1.34b003beee67bp-4}, {0x1.d40cc001e9fd6p-1, 0x1.6f9d9b2997907p-4},
  {0x1.cd755ceee8d1dp-1, 0x1.a9b53d561f1bbp-4}, {0x1.c70cd772d5babp-1, 0x1.e2fce6417493fp-4},
  {0x1.c0d1429125649p-1, 0x1.0dbd2942908b4p-3}, {0x1.bac0cbf246f1cp-1, 0x1.299981652da3dp-3},
  {0x1.b4d9ba1cf7947p-1, 0x1.45161f80ebbedp-3}, {0x1.af1a6ad37a2eep-1, 0x1.60358cb284e6ep-3},
  {0x1.a98151916fce0p-1, 0x1.7afa38a290ab0p-3}, {0x1.a40cf6276378ap-1, 0x1.95667ad5d0993p-3},
  {0x1.9ebbf37167b25p-1, 0x1.af7c93e811547p-3}, {0x1.998cf626676a3p-1, 0x1.c93eaeb33e210p-3},
  {0x1.947ebbbe07efcp-1, 0x1.e2aee16420b79p-3}, {0x1.8f90116b2ed26p-1, 0x1.fbcf2e7e26c9fp-3},
  {0x1.8abfd3296e17dp-1, 0x1.0a50c2e7b4a5bp-2}, {0x1.860ceadbc3a2fp-1, 0x1.1693e2ab0b8b5p-2},
  {0x1.81764f7b3e95dp-1, 0x1.22b1dd0c26444p-2}, {0x1.7cfb04543f712p-1, 0x1.2eab9077a5f5bp-2},
  {0x1.789a185126283p-1, 0x1.3a81d3a37b68fp-2}, {0x1.7452a5515cb7bp-1, 0x1.463575e92b4a2p-2},
  {0x1.7023cf8bc561ap-1, 0x1.51c73f9af8e78p-2}, {0x1.6c0cc4fba9cedp-1, 0x1.5d37f2544ef66p-2}};
static inline double pm_log(double x)
{
  const uint64_t b = d2u(x);
  const uint64_t tmp = b - 0x3FE6A09E00000000ULL;
  const int i = (int)((tmp >> 46) & 63);
  const int64_t k = (int64_t)tmp >> 52;
  const double z = u2d(b - (tmp & 0xFFF0000000000000ULL));
  const double r = fma(z, PM_LOG_T[i][0], -1.0);
  double q = -0.125;
  q = fma(q, r, 0.14285714285714285);            /*  1/7 */
  q = fma(q, r, -0.16666666666666666);           /* -1/6 */
  q = fma(q, r, 0.20000000000000001);            /*  1/5 */
  q = fma(q, r, -0.25);
  q = fma(q, r, 0.33333333333333331);            /*  1/3 */
  q = fma(q, r, -0.5);
  const double hi = fma((double)k, 0.69314718055994529, PM_LOG_T[i][1]);
  return hi + fma(r * r, q, r);
}

static inline double pm_exp(double x)
{
  if (x < -745.0) return 0.0;
  if (x > 709.0) return INFINITY;
  double kf = floor(x * 1.4426950408889634 + 0.5);
  double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
  double p = 1.6059043836821613e-10;          /* 1/13! */
  p = p * r + 2.08767569878681e-09;           /* 1/12! */
  p = p * r + 2.505210838544172e-08;          /* 1/11! */
  p = p * r + 2.755731922398589e-07;          /* 1/10! */
  p = p * r + 2.7557319223985893e-06;         /* 1/9!  */
  p = p * r + 2.48015873015873e-05;           /* 1/8!  */
  p = p * r + 0.0001984126984126984;          /* 1/7!  */
  p = p * r + 0.001388888888888889;           /* 1/6!  */
  p = p * r + 0.008333333333333333;           /* 1/5!  */
  p = p * r + 0.041666666666666664;           /* 1/4!  */
  p = p * r + 0.16666666666666666;            /* 1/3!  */
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  int k = (int)kf;
  /* scale by 2^k in two exact steps so that results in the double-denormal range stay defined */
  int k1 = k / 2, k2 = k - k1;
  double s1 = u2d((uint64_t)(k1 + 1023) << 52), s2 = u2d((uint64_t)(k2 + 1023) << 52);
  return (p * s1) * s2;
}

/* expf of the portable mode: the single-precision exp algorithm of the C library the reference's CPU build calls (glibc
 * 2.35 sysdeps/ieee754/flt-32/e_expf.c, i.e. the ARM "optimized routines" expf: exp(x) = 2^(k/32) 2^(r/32) with
 * k + r = 32 x / ln2, |r| <= 1/2, a 32-entry table for the first factor and a cubic for the second, all in double, ONE
 * rounding to float at the end), written in the form x86-64 machines with FMA execute it (every multiply-add one fused
 * operation, spelled fma() here).  The table is generated by oracle/gen_exp2f_table.py (pure arithmetic); the scalar
 * constants are those of the published algorithm.  oracle/check_libm.c compares the function with the host libm's expf()
 * over all floats: 0 of 4 278 190 082 differ on the build machine (tests/test_oracle_golden.py repeats a sampled
 * comparison).  It replaces the degree-13 Taylor polynomial this mode used for expf before (170 648 floats differed from
 * libm), at a third of the operations -- expf is called twice per electron shell in every Compton angle trial.
 * logf, powf, sin and cos of this mode remain the pm_ functions above / below. */
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static const uint64_t GL_EXP2_T[32] = {
  0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
  0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
  0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
  0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
  0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
  0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
static inline float gl_expf(float x)
{
  const uint32_t abstop = (f2u(x) >> 20) & 0x7ff;
  if (abstop >= (0x42b00000u >> 20)) {            /* |x| >= 88, inf or nan */
    if (f2u(x) == 0xff800000u) return 0.0f;
    if (abstop >= (0x7f800000u >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return INFINITY;       /* x > log(0x1p128) */
    if (x < -0x1.9fe368p6f) return 0.0f;          /* x < log(0x1p-150) */
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;     /* x < log(0x1p-149): the library's "may underflow" value */
  }
  const double xd = (double)x;
  double kd = fma(0x1.71547652b82fep+5, xd, 0x1.8p+52);          /* 32/ln2 * x, rounded to an integer by the shift */
  const uint64_t ki = d2u(kd);
  kd -= 0x1.8p+52;
  const double r = fma(0x1.71547652b82fep+5, xd, -kd);
  const double s = u2d(GL_EXP2_T[ki % 32] + (ki << 47));          /* 2^(k/32) */
  const double z = fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = fma(0x1.62e42ff0c52d6p-6, r, 1.0);
  y = fma(z, r2, y);
  return (float)(y * s);
}

static inline void pm_sincos(double x, double *sn, double *cs)
{
  double kf = floor(x * 0.63661977236758138 + 0.5);
  int q = ((int)kf) & 3;
  double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
  double z = r * r;
  double ps = -8.2206352466243295e-18;        /* -1/19! */
  ps = ps * z + 2.8114572543455206e-15;       /*  1/17! */
  ps = ps * z - 7.6471637318198164e-13;       /* -1/15! */
  ps = ps * z + 1.6059043836821613e-10;       /*  1/13! */
  ps = ps * z - 2.505210838544172e-08;        /* -1/11! */
  ps = ps * z + 2.7557319223985893e-06;       /*  1/9!  */
  ps = ps * z - 0.0001984126984126984;        /* -1/7!  */
  ps = ps * z + 0.008333333333333333;         /*  1/5!  */
  ps = ps * z - 0.16666666666666666;          /* -1/3!  */
  double s = r + r * (z * ps);
  double pc = 4.1103176233121648e-19;         /*  1/20! */
  pc = pc * z - 1.5619206968586225e-16;       /* -1/18! */
  pc = pc * z + 4.7794773323873853e-14;       /*  1/16! */
  pc = pc * z - 1.1470745597729725e-11;       /* -1/14! */
  pc = pc * z + 2.08767569878681e-09;         /*  1/12! */
  pc = pc * z - 2.755731922398589e-07;        /* -1/10! */
  pc = pc * z + 2.48015873015873e-05;         /*  1/8!  */
  pc = pc * z - 0.001388888888888889;         /* -1/6!  */
  pc = pc * z + 0.041666666666666664;         /*  1/4!  */
  pc = pc * z - 0.5;
  double c = 1.0 + z * pc;
  if (q == 0) { *sn = s; *cs = c; }
  else if (q == 1) { *sn = c; *cs = -s; }
  else if (q == 2) { *sn = -s; *cs = -c; }
  else { *sn = -c; *cs = s; }
}

double oracle_pm_log(double x) { return pm_log(x); }
double oracle_pm_exp(double x) { return pm_exp(x); }
float oracle_gl_expf(float x) { return gl_expf(x); }
void oracle_pm_sincos(double x, double *s, double *c) { pm_sincos(x, s, c); }

/* math dispatch ---------------------------------------------------------------------------- */
static inline float m_logf(float x, int pm) { return pm ? (float)pm_log((double)x) : logf(x); }
static inline float m_expf(float x, int pm) { return pm ? gl_expf(x) : expf(x); }
static inline float m_powf(float b, float y, int pm) { return pm ? (float)pm_exp((double)y * pm_log((double)b)) : powf(b, y); }
static inline void m_sincos(double x, double *s, double *c, int pm)
{ if (pm) pm_sincos(x, s, c); else { *s = sin(x); *c = cos(x); } }

/* ------------------------------------------------------------------------------------------
 * RANECU  (K.cu:965-1015 ranecu/ranecu_double, :919-950 abMODm, :841-894 init_PRNG)
 * ------------------------------------------------------------------------------------------ */
static inline int ranecu_step(i2 *s)
{
  int i1 = s->x / 53668;
  s->x = 40014 * (s->x - i1 * 53668) - i1 * 12211;
  int k = s->y / 52774;
  s->y = 40692 * (s->y - k * 52774) - k * 3791;
  if (s->x < 0) s->x += 2147483563;
  if (s->y < 0) s->y += 2147483399;
  k = s->x - s->y;
  if (k < 1) k += 2147483562;
  return k;
}
static inline float ranecu(i2 *s, uint64_t *cnt) { ++*cnt; return (float)ranecu_step(s) * 4.65661305739e-10f; }
static inline double ranecu_d(i2 *s, uint64_t *cnt) { ++*cnt; return (double)ranecu_step(s) * 4.6566130573917692e-10; }

static int abmodm(int m, int a, int s)
{
  int q, k, p = -m;
  while (a > 32768) {
    if (a & 1) { p += s; if (p > 0) p -= m; }
    a >>= 1;
    s = (s - m) + s;
    if (s < 0) s += m;
  }
  q = m / a;
  k = s / q;
  s = a * (s - k * q) - k * (m - q * a);
  while (s < 0) s += m;
  p += s;
  if (p < 0) p += m;
  return p;
}
static int pow_mod(int m, int a, unsigned long long leap)
{
  int y = 1, z = a;
  for (;;) {
    if (leap & 1ULL) { leap >>= 1; y = abmodm(m, z, y); if (leap == 0) break; }
    else leap >>= 1;
    z = abmodm(m, z, z);
  }
  return y;
}
static void init_prng(int batch, int hpt, int seed_input, i2 *seed)
{
  unsigned long long leap = ((unsigned long long)(batch + 1)) * (unsigned long long)(hpt * 256);
  seed->x = abmodm(2147483563, seed_input, pow_mod(2147483563, 40014, leap));
  seed->y = abmodm(2147483399, seed_input, pow_mod(2147483399, 40692, leap));
}
void oracle_init_prng(int batch, int hpt, int seed_input, int *seed2)
{ i2 s; init_prng(batch, hpt, seed_input, &s); seed2[0] = s.x; seed2[1] = s.y; }
float oracle_ranecu(int *seed2)
{ i2 s = { seed2[0], seed2[1] }; uint64_t c = 0; float r = ranecu(&s, &c); seed2[0] = s.x; seed2[1] = s.y; return r; }
double oracle_ranecu_double(int *seed2)
{ i2 s = { seed2[0], seed2[1] }; uint64_t c = 0; double r = ranecu_d(&s, &c); seed2[0] = s.x; seed2[1] = s.y; return r; }
int oracle_abmodm(int m, int a, int s) { return abmodm(m, a, s); }
/* update_seed_PRNG, MC-GPU_v1.3.cu:3456-3485 */
int oracle_update_seed(int batch_number, unsigned long long total_histories, int seed)
{
  if (batch_number == 0) return seed;
  return abmodm(2147483563, seed, pow_mod(2147483563, 40014, total_histories * (unsigned long long)(batch_number * 256)));
}

/* ------------------------------------------------------------------------------------------
 * Geometry helpers: move_to_bbox (K.cu:714-805), locate_voxel (K.cu:1033-1065)
 * ------------------------------------------------------------------------------------------ */
static inline float entry_dist(float p, float d, float size)
{
  if (d > EPS_SRC) return (p > 0.0f) ? 0.0f : EPS_SRC + (-p) / d;
  if (d < -EPS_SRC) return (p < size) ? 0.0f : EPS_SRC + (size - p) / d;
  return NEG_INF_DIST;
}
static void move_to_bbox(f3 *pos, const f3 *dir, const float *bbox, int *flag)
{
  float dy = entry_dist(pos->y, dir->y, bbox[1]);
  float dx = entry_dist(pos->x, dir->x, bbox[0]);
  float dz = entry_dist(pos->z, dir->z, bbox[2]);
  if ((dy > dx) && (dy > dz)) dz = dy;
  else if (dx > dz) dz = dx;
  pos->x += dz * dir->x; pos->y += dz * dir->y; pos->z += dz * dir->z;
  if ((pos->x < 0.0f) || (pos->x > bbox[0]) || (pos->y < 0.0f) || (pos->y > bbox[1]) || (pos->z < 0.0f) || (pos->z > bbox[2])) {
    pos->x -= dz * dir->x; pos->y -= dz * dir->y; pos->z -= dz * dir->z;
    *flag = -111;
  }
}
static inline int locate_voxel(const oracle_tables *T, const f3 *p)
{
  if ((p->y < EPS_SRC) || (p->y > (T->size_bbox[1] - EPS_SRC)) || (p->x < EPS_SRC) || (p->x > (T->size_bbox[0] - EPS_SRC)) ||
      (p->z < EPS_SRC) || (p->z > (T->size_bbox[2] - EPS_SRC)))
    return -1;
  int ix = (int)(p->x * T->inv_voxel_size[0]);
  int iy = (int)(p->y * T->inv_voxel_size[1]);
  int iz = (int)(p->z * T->inv_voxel_size[2]);
  return ix + iy * T->num_voxels[0] + iz * T->num_voxels[0] * T->num_voxels[1];
}

/* ------------------------------------------------------------------------------------------
 * source (K.cu:626-686): Walker-alias energy + rejection-sampled direction in the fan, rotate, enter bbox
 * ------------------------------------------------------------------------------------------ */
static void source(const oracle_tables *T, const src_t *S, const det_t *D, f3 *pos, f3 *dir, float *energy, i2 *seed, int *absvox,
                   int pm, uint64_t *cnt)
{
  float RN = ranecu(seed, cnt) * T->num_bins_espc;
  int ip = (int)RN;
  float fr = RN - (float)ip;
  int bin = (fr < T->espc_cutoff[ip]) ? ip : (int)T->espc_alias[ip];
  *energy = T->espc[bin] + ranecu(seed, cnt) * (T->espc[bin + 1] - T->espc[bin]);
  do {
    dir->z = S->cos_theta_low + ranecu(seed, cnt) * S->D_cos_theta;
    float phi = S->phi_low + ranecu(seed, cnt) * S->D_phi;
    float sth = sqrtf(1.0f - dir->z * dir->z);
    double sd, cd;
    m_sincos((double)phi, &sd, &cd, pm);
    float sphi = (float)sd, cphi = (float)cd;
    dir->y = sth * sphi;
    dir->x = sth * cphi;
  } while (fabsf(dir->z / (dir->y + 1.0e-7f)) > S->max_height_at_y1cm);
  if (D->rotation_flag == 1) {
    float tx = dir->x, ty = dir->y;
    dir->x = S->rot_fan[0] * tx + S->rot_fan[1] * ty + S->rot_fan[2] * dir->z;
    dir->y = S->rot_fan[3] * tx + S->rot_fan[4] * ty + S->rot_fan[5] * dir->z;
    dir->z = S->rot_fan[6] * tx + S->rot_fan[7] * ty + S->rot_fan[8] * dir->z;
  }
  *pos = S->position;
  move_to_bbox(pos, dir, T->size_bbox, absvox);
}

/* ------------------------------------------------------------------------------------------
 * rotate_double (K.cu:1103-1148, PENELOPE DIRECT)
 * ------------------------------------------------------------------------------------------ */
static void rotate_dir(f3 *d, double costh, double phi, int pm)
{
  double DXY, NORM, cosphi, sinphi, SDT;
  DXY = d->x * d->x + d->y * d->y;             /* float arithmetic, then widened (as in the reference) */
  m_sincos(phi, &sinphi, &cosphi, pm);
  NORM = DXY + d->z * d->z;
  if (fabs(NORM - 1.0) > 1.0e-14) {
    NORM = 1.0 / sqrt(NORM);
    d->x = NORM * d->x; d->y = NORM * d->y; d->z = NORM * d->z;
    DXY = d->x * d->x + d->y * d->y;
  }
  if (DXY > 1.0e-28) {
    SDT = sqrt((1.0 - costh * costh) / DXY);
    float xin = d->x;
    d->x = d->x * costh + SDT * (xin * d->z * cosphi - d->y * sinphi);
    d->y = d->y * costh + SDT * (d->y * d->z * cosphi + xin * sinphi);
    d->z = d->z * costh - DXY * SDT * cosphi;
  } else {
    SDT = sqrt(1.0 - costh * costh);
    d->y = SDT * sinphi;
    if (d->z > 0.0) { d->x = SDT * cosphi; d->z = costh; }
    else { d->x = -SDT * cosphi; d->z = -costh; }
  }
}
void oracle_rotate(float *dir3, double costh, double phi, int math_mode)
{ f3 d = { dir3[0], dir3[1], dir3[2] }; rotate_dir(&d, costh, phi, math_mode); dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z; }

/* ------------------------------------------------------------------------------------------
 * GRAa (K.cu:1181-1246): Rayleigh angle by RITA sampling of the squared form factor
 * ------------------------------------------------------------------------------------------ */
static void graa(const oracle_tables *T, float energy, double *costh, int mat, float pmax_current, i2 *seed, uint64_t *cnt)
{
  double xmax = ((double)energy) * 8.065535669099010e-5;
  double xl = (double)T->xco[(mat + 1) * NPRAY - 1];
  double x2max = (xmax * xmax < xl) ? xmax * xmax : xl;
  if (xmax < 0.01) {
    do { *costh = 1.0 - ranecu_d(seed, cnt) * 2.0; } while (ranecu_d(seed, cnt) > (((*costh) * (*costh) + 1.0) * 0.5));
    return;
  }
  for (;;) {
    double ru = ranecu_d(seed, cnt) * (double)pmax_current;
    int itn = (int)(ru * (NPRAY - 1));
    int i = (int)T->itlco[itn + mat * NPRAY];
    int j = (int)T->ituco[itn + mat * NPRAY];
    if ((j - i) > 1) {
      do {
        int k = (i + j) >> 1;
        if (ru > T->pco[k - 1 + mat * NPRAY]) i = k; else j = k;
      } while ((j - i) > 1);
    }
    int q = i - 1 + mat * NPRAY;
    double rr = ru - T->pco[q];
    double xx;
    if (rr > 1e-16) {
      double d = (double)(T->pco[q + 1] - T->pco[q]);
      float a = T->aco[q], b = T->bco[q], x0 = T->xco[q];
      xx = (double)x0 + (double)(a + 1.0f + b) * d * rr / (d * d + (a * d + b * rr) * rr) * (double)(T->xco[q + 1] - x0);
    } else {
      xx = T->xco[q];
    }
    if (xx < x2max) {
      *costh = 1.0 - 2.0 * xx / x2max;
      if (ranecu_d(seed, cnt) < (((*costh) * (*costh) + 1.0) * 0.5)) break;
    }
  }
}
void oracle_graa(const oracle_tables *T, float energy, double *costh, int mat, int index, int *seed2)
{ i2 s = { seed2[0], seed2[1] }; uint64_t c = 0; graa(T, energy, costh, mat, T->pmax[(index + 1) * MAXMAT + mat], &s, &c); seed2[0] = s.x; seed2[1] = s.y; }

/* ------------------------------------------------------------------------------------------
 * GCOa (K.cu:1287-1515): Compton, impulse approximation with analytical one-electron profiles.
 * rn[] is zero-initialised here; the reference leaves entries of inactive shells (U_i >= E)
 * uninitialised -- unreachable for the bundled materials (max U_i = 4.04 keV < 5 keV cut-off).
 * ------------------------------------------------------------------------------------------ */
static inline float shell_pz(float fj0, float aux, float u)
{ return fj0 * (aux - u * 510998.918f) / (sqrtf(aux + aux + u * u) * 510998.918f); }

static void gcoa(const oracle_tables *T, float *energy, double *costh, int mat, i2 *seed, int pm, uint64_t *cnt)
{
  float s, a1, s0, af, ek, ek2, ek3, tau, pzomc = 0.0f, taumin;
  float rn[MAXSHELLS];
  double cdt1;
  int i, nosc = T->noscco[mat];
  const float E = *energy;
  memset(rn, 0, sizeof rn);
  ek = E * 1.956951306108245e-6f;
  ek2 = ek * 2.f + 1.f;
  ek3 = ek * ek;
  taumin = 1.f / ek2;
  a1 = m_logf(ek2, pm);
  s0 = 0.0f;
  for (i = 0; i < nosc; i++) {
    float t = T->uico[mat + i * MAXMAT];
    if (t < E) {
      float aux = E * (E - t) * 2.f;
      pzomc = shell_pz(T->fj0[mat + i * MAXMAT], aux, t);
      if (pzomc > 0.0f) t = (0.707106781186545f + pzomc * 1.4142135623731f) * (0.707106781186545f + pzomc * 1.4142135623731f);
      else t = (0.707106781186545f - pzomc * 1.4142135623731f) * (0.707106781186545f - pzomc * 1.4142135623731f);
      t = 0.5f * m_expf(0.5f - t, pm);
      if (pzomc > 0.0f) t = 1.0f - t;
      s0 += T->fco[mat + i * MAXMAT] * t;
    }
  }
  do {
    if (ranecu(seed, cnt) * (a1 + 2. * ek * (ek + 1.f) * taumin * taumin) < a1)
      tau = m_powf(taumin, ranecu(seed, cnt), pm);
    else
      tau = sqrtf(1.f + ranecu(seed, cnt) * (taumin * taumin - 1.f));
    cdt1 = (double)(1.f - tau) / (((double)tau) * ((double)E) * 1.956951306108245e-6);
    if (cdt1 > 2.0) cdt1 = 1.99999999;
    s = 0.0f;
    for (i = 0; i < nosc; i++) {
      float t = T->uico[mat + i * MAXMAT];
      if (t < E) {
        float aux = E * (E - t) * ((float)cdt1);
        if ((aux > 1.0e-12f) || (t > 1.0e-12f)) pzomc = shell_pz(T->fj0[mat + i * MAXMAT], aux, t);
        else pzomc = 0.002f;
        t = pzomc * 1.4142135623731f;
        if (pzomc > 0.0f) t = 0.5f - (t + 0.70710678118654502f) * (t + 0.70710678118654502f);
        else t = 0.5f - (0.70710678118654502f - t) * (0.70710678118654502f - t);
        t = 0.5f * m_expf(t, pm);
        if (pzomc > 0.0f) t = 1.0f - t;
        s += T->fco[mat + i * MAXMAT] * t;
        rn[i] = t;
      }
    }
  } while ((ranecu(seed, cnt) * s0) > (s * (1.0f + tau * ((ek3 - ek2 - 1.0f) + tau * (ek2 + tau * ek3))) / (ek3 * tau * (tau * tau + 1.0f))));
  *costh = 1.0 - cdt1;
  for (;;) {
    float t = s * ranecu(seed, cnt);
    float pac = 0.0f;
    int ishell = nosc - 1;
    for (i = 0; i < (nosc - 1); i++) {
      pac += T->fco[mat + i * MAXMAT] * rn[i];
      if (pac > t) { ishell = i; break; }
    }
    t = ranecu(seed, cnt) * rn[ishell];
    if (t < 0.5f) pzomc = (0.70710678118654502f - sqrtf(0.5f - m_logf(t + t, pm))) / (T->fj0[mat + ishell * MAXMAT] * 1.4142135623731f);
    else pzomc = (sqrtf(0.5f - m_logf(2.0f - 2.0f * t, pm)) - 0.70710678118654502f) / (T->fj0[mat + ishell * MAXMAT] * 1.4142135623731f);
    if (pzomc < -1.0f) continue;
    t = tau * (tau - (*costh) * 2.f) + 1.f;      /* double expression rounded to float on assignment */
    if (t > 1.0e-20f) af = sqrtf(t) * (tau * (tau - ((float)(*costh))) / t + 1.f);
    else af = 0.00200f;
    if (af > 0.0f) t = af * 0.2f + 1.f; else t = 1.f - af * 0.2f;
    {
      float pzc = (pzomc < 0.2f) ? pzomc : 0.2f;
      pzc = (pzc > -0.2f) ? pzc : -0.2f;
      if (ranecu(seed, cnt) * t < (af * pzc + 1.f)) break;
    }
  }
  {
    float t = pzomc * pzomc;
    float b1 = 1.f - t * tau * tau;
    float b2 = 1.f - t * tau * ((float)(*costh));
    float r = sqrtf(fabsf(b2 * b2 - b1 * (1.0f - t)));
    if (pzomc < 0.0f) r *= -1.0f;
    t = (tau / b1) * (b2 + r);
    if (t > 1.0f) t = 1.0f;
    *energy = E * t;
  }
}
/* S(E, theta) = sum_i f_i n_i(E, theta) in the reference's float arithmetic (K.cu:1300-1314 for cdt = 2, i.e. S0; :1340-1366 for
 * the angle of a trial): what the product's COMPAT kernel brackets with its S0 bounds (tests/test_host_tables.py) */
float oracle_compton_s(const oracle_tables *T, float E, float cdt, int mat, int math_mode)
{
  float s = 0.0f;
  for (int i = 0; i < T->noscco[mat]; i++) {
    float t = T->uico[mat + i * MAXMAT];
    if (t < E) {
      float aux = E * (E - t) * cdt, pz;
      if ((aux > 1.0e-12f) || (t > 1.0e-12f)) pz = shell_pz(T->fj0[mat + i * MAXMAT], aux, t);
      else pz = 0.002f;
      t = pz * 1.4142135623731f;
      if (pz > 0.0f) t = 0.5f - (t + 0.70710678118654502f) * (t + 0.70710678118654502f);
      else t = 0.5f - (0.70710678118654502f - t) * (0.70710678118654502f - t);
      t = 0.5f * m_expf(t, math_mode);
      if (pz > 0.0f) t = 1.0f - t;
      s += T->fco[mat + i * MAXMAT] * t;
    }
  }
  return s;
}
void oracle_gcoa(const oracle_tables *T, float *energy, double *costh, int mat, int *seed2, int math_mode)
{ i2 s = { seed2[0], seed2[1] }; uint64_t c = 0; gcoa(T, energy, costh, mat, &s, math_mode, &c); seed2[0] = s.x; seed2[1] = s.y; }
void oracle_source(const oracle_tables *T, int num_p, int *seed2, float *pos3, float *dir3, float *energy, int *absvox, int math_mode)
{
  i2 s = { seed2[0], seed2[1] }; uint64_t c = 0; f3 p, d;
  *absvox = 1;
  source(T, (const src_t *)T->source_data + num_p, (const det_t *)T->detector_data + num_p, &p, &d, energy, &s, absvox, math_mode, &c);
  pos3[0] = p.x; pos3[1] = p.y; pos3[2] = p.z; dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z; seed2[0] = s.x; seed2[1] = s.y;
}

/* ------------------------------------------------------------------------------------------
 * tally_image (K.cu:482-604), CPU branches (:553-565 rotated detector, :589-600 detector at +Y)
 * ------------------------------------------------------------------------------------------ */
static inline void tally_add(uint64_t *word, unsigned long long v, int shared)
{
  if (shared) __atomic_fetch_add(word, (uint64_t)v, __ATOMIC_RELAXED);  /* OpenMP runs: integer sum, order-independent */
  else *word += (uint64_t)v;
}

static inline void tally_pixel(uint64_t *image, uint64_t *w2, int word, float energy, int shared)
{
  const unsigned long long w = (unsigned long long)(energy * 100.0f + 0.5f);
  tally_add(&image[word], w, shared);
  if (w2) tally_add(&w2[word], (w >> 10) * (w >> 10), shared);
}

static void tally_image(float energy, f3 *pos, const f3 *dir, int scatter_state, uint64_t *image, uint64_t *w2, const src_t *S, const det_t *D,
                        oracle_counters *C, int shared)
{
  float dist, rot;
  C->tally_calls++;
  if (D->rotation_flag == 1) {
    float cosang = dir->x * S->direction.x + (dir->y * S->direction.y + (dir->z * S->direction.z));
    if (cosang < 0.025f) return;
    dist = (S->direction.x * (D->center.x - pos->x) + (S->direction.y * (D->center.y - pos->y) + (S->direction.z * (D->center.z - pos->z)))) / cosang;
    pos->x = pos->x + dist * dir->x;
    pos->y = pos->y + dist * dir->y;
    pos->z = pos->z + dist * dir->z;
    rot = D->rot_inv[0] * pos->x + D->rot_inv[1] * pos->y + D->rot_inv[2] * pos->z;
    float px = floor((rot - D->corner_min_rotated_to_Y.x) * D->inv_pixel_size_X);
    if ((px > -0.1f) && (px < (D->num_pixels_x - 0.1f))) {
      rot = D->rot_inv[6] * pos->x + D->rot_inv[7] * pos->y + D->rot_inv[8] * pos->z;
      float pz = floor((rot - D->corner_min_rotated_to_Y.z) * D->inv_pixel_size_Z);
      if ((pz > -0.1f) && (pz < (D->num_pixels_y - 0.1f))) {
        tally_pixel(image, w2, (int)(((float)scatter_state) * D->total_num_pixels + px + pz * D->num_pixels_x + 0.0001f), energy, shared);
        C->tally_hits++;
      }
    }
  } else {
    if (dir->y < 0.0001f) return;
    dist = (D->center.y - pos->y) / (dir->y);
    float px = floor((pos->x + dist * dir->x - D->corner_min_rotated_to_Y.x) * D->inv_pixel_size_X);
    if ((px > -0.1f) && (px < (D->num_pixels_x - 0.1f))) {
      float pz = floor((pos->z + dist * dir->z - D->corner_min_rotated_to_Y.z) * D->inv_pixel_size_Z);
      if ((pz > -0.1f) && (pz < (D->num_pixels_y - 0.1f))) {
        tally_pixel(image, w2, (int)(((float)scatter_state) * D->total_num_pixels + px + pz * D->num_pixels_x + 0.0001f), energy, shared);
        C->tally_hits++;
      }
    }
  }
}

/* tally_materials_dose (K.cu:1547-1563) + tally_voxel_energy_deposition (K.cu:418-443), CPU-build rounding */
static void tally_dose(const oracle_tables *T, float Edep, int mat, const f3 *pos, int shared)
{
  if (T->materials_dose) {
    tally_add(&T->materials_dose[2 * mat], (unsigned long long)(Edep * 100.0f + 0.5f), shared);
    tally_add(&T->materials_dose[2 * mat + 1], (unsigned long long)(Edep * Edep + 0.5f), shared);
  }
  if (T->voxels_edep && T->dose_roi[1] > -1) {
    const int *r = T->dose_roi;
    const int x = (short)(int)(pos->x * T->inv_voxel_size[0]), y = (short)(int)(pos->y * T->inv_voxel_size[1]),
              z = (short)(int)(pos->z * T->inv_voxel_size[2]);
    if (x < r[0] || x > r[1] || y < r[2] || y > r[3] || z < r[4] || z > r[5]) return;
    {
      const int DX = 1 + (r[1] - r[0]);
      const int v = (x - r[0]) + (y - r[2]) * DX + (z - r[4]) * DX * (1 + (r[3] - r[2]));
      tally_add(&T->voxels_edep[2 * (size_t)v], (unsigned long long)(Edep * 100.0f + 0.5f), shared);
      tally_add(&T->voxels_edep[2 * (size_t)v + 1], (unsigned long long)(Edep * Edep + 0.5f), shared);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * track_particles (K.cu:120-384): one batch of `hpt` histories
 * ------------------------------------------------------------------------------------------ */
static void track_batch(const oracle_tables *T, int batch, int hpt, int num_p, int seed_input, uint64_t *image, int pm, oracle_counters *C, int shared)
{
  const src_t *S = (const src_t *)T->source_data + num_p;
  const det_t *D = (const det_t *)T->detector_data + num_p;
  const float *vox = T->voxel_mat_dens, *Wt = T->mfp_woodcock, *A = T->mfp_a, *B = T->mfp_b;
  f3 pos, dir;
  float energy, step, prob, randno, mfp_density, mfpW;
  float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
  i2 seed;
  int index, mat = 0, mat_old, scatter_state;
  init_prng(batch, hpt, seed_input, &seed);
  for (; hpt > 0; hpt--) {
    int absvox = 1;
    C->histories++;
    source(T, S, D, &pos, &dir, &energy, &seed, &absvox, pm, &C->rng);
    scatter_state = 0;
    index = (int)((energy - T->e0) * T->ide + 0.00001f);
    mfpW = Wt[2 * index] + energy * Wt[2 * index + 1];
    C->woodcock_reads++;
    mat_old = -1;
    for (;;) {
      float dens = 0.f;
      if (absvox < 0) break;
      do {
        step = -(mfpW)*m_logf(ranecu(&seed, &C->rng), pm);
        C->steps++;
        pos.x += step * dir.x;
        pos.y += step * dir.y;
        pos.z += step * dir.z;
        absvox = locate_voxel(T, &pos);
        if (absvox < 0) break;
        C->voxel_reads++;
        mat = (int)(vox[2 * (size_t)absvox] - 1);
        dens = vox[2 * (size_t)absvox + 1];
        if (mat != mat_old) {
          const float *a = A + 3 * ((size_t)index * MAXMAT + mat), *b = B + 3 * ((size_t)index * MAXMAT + mat);
          ax = a[0]; ay = a[1]; az = a[2]; bx = b[0]; by = b[1]; bz = b[2];
          mat_old = mat;
          C->mfp_reads++;
        }
        mfp_density = mfpW * dens;
        prob = 1.0f - mfp_density * (ax + energy * bx);
        randno = ranecu(&seed, &C->rng);
      } while (randno < prob);
      if (absvox < 0) break;
      prob += mfp_density * (ay + energy * by);
      if (randno < prob) {
        double costh;
        C->compton++;
        randno = energy;
        gcoa(T, &energy, &costh, mat, &seed, pm, &C->rng);
        rotate_dir(&dir, costh, 6.28318530717958647693 * ranecu_d(&seed, &C->rng), pm);
        randno = energy - randno;  /* minus the energy lost (K.cu:299) */
        index = (int)((energy - T->e0) * T->ide + 0.00001f);
        if (index > -1) {
          mfpW = Wt[2 * index] + energy * Wt[2 * index + 1];
          C->woodcock_reads++;
          mat_old = -2;
          scatter_state = (scatter_state == 0) ? 1 : 3;
        }
      } else {
        prob += mfp_density * (az + energy * bz);
        if (randno < prob) {
          double costh;
          C->rayleigh++;
          graa(T, energy, &costh, mat, T->pmax[(index + 1) * MAXMAT + mat], &seed, &C->rng);
          rotate_dir(&dir, costh, 6.28318530717958647693 * ranecu_d(&seed, &C->rng), pm);
          scatter_state = (scatter_state == 0) ? 2 : 3;
        } else {
          C->photo++;
          randno = -energy;
          index = -11;
        }
      }
      if (randno < -0.001f) tally_dose(T, -1.0f * randno, mat, &pos, shared);  /* K.cu:356-367 */
      if (index < 0) break;
    }
    if (index > -1) tally_image(energy, &pos, &dir, scatter_state, image, T->image_w2, S, D, C, shared);
  }
}

static void add_counters(oracle_counters *a, const oracle_counters *b)
{
  a->histories += b->histories; a->steps += b->steps; a->voxel_reads += b->voxel_reads; a->mfp_reads += b->mfp_reads;
  a->woodcock_reads += b->woodcock_reads; a->compton += b->compton; a->rayleigh += b->rayleigh; a->photo += b->photo;
  a->rng += b->rng; a->tally_calls += b->tally_calls; a->tally_hits += b->tally_hits;
}

int oracle_track(const oracle_tables *T, int num_p, int seed_input, int batch0, int nbatches, int hpt, uint64_t *image, int math_mode,
                 int n_threads, oracle_counters *counters)
{
  oracle_counters total;
  memset(&total, 0, sizeof total);
  if (n_threads <= 1) {
    int b;
    for (b = batch0; b < batch0 + nbatches; b++) track_batch(T, b, hpt, num_p, seed_input, image, math_mode, &total, 0);
  } else {
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
    {
      oracle_counters c;
      int b;
      memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 4)
      for (b = batch0; b < batch0 + nbatches; b++) track_batch(T, b, hpt, num_p, seed_input, image, math_mode, &c, 1);
#pragma omp critical
      add_counters(&total, &c);
    }
#else
    int b;
    for (b = batch0; b < batch0 + nbatches; b++) track_batch(T, b, hpt, num_p, seed_input, image, math_mode, &total, 0);
#endif
  }
  if (counters) add_counters(counters, &total);
  return 0;
}
