/* rng_battery.c -- TEST INFRASTRUCTURE (never linked into the product): a TestU01-style battery for the FAST personality's
 * per-history random-number streams (4d-cbct-mc_amd/csrc/track_common.inc: rng_init_history / rng_u32; restated here in plain
 * C like oracle/fast_rng.py restates them in numpy -- tests/test_fast_rng.py holds the two restatements to the same words).
 *
 * What is tested is not one long stream (a history draws a few dozen numbers) but what a projection consumes: 1e8 SHORT streams,
 * seeded from consecutive history ids.  The serial sequence handed to the tests is built from ids x draws in two orders:
 *     history-major   id 0: draws 0..31, id 1: draws 0..31, ...      (what one lane sees)
 *     draw-major      draw k of ids 0..2^20-1, then draw k+1 ...     (neighbouring ids at the same depth: what a counter-seeded
 *                                                                     family of streams could get wrong)
 * for projections 0 and 893 (the counter word the kernel calls stream_key), on three bit selections: all 32 bits (the alias draws
 * and rng_d use them), the 24 upper bits (rng_f) and the 8 low bits (discarded by rng_f).
 * The same battery runs on the yardstick of tests/test_fast_rng.py: Philox4x32-10 evaluated PER DRAW (Salmon et al., SC11).
 *
 * Tests (null distributions as in Knuth TAOCP 2 3.3.2, L'Ecuyer & Simard's TestU01 guide, NIST SP 800-22):
 *     birthday spacings   n points in k = 2^(t b) cells, collisions among the sorted spacings ~ Poisson(n^3 / 4k)
 *     collisions          n balls in k = 2^40 urns, C = n - distinct, against its exact mean and Poisson variance
 *     matrix rank         b x b matrices over GF(2) from b consecutive values, chi^2 on {<= b-2, b-1, b}
 *     linear complexity   Berlekamp-Massey on blocks of 1000 bits of ONE bit position (0, 8, 31), NIST's seven classes
 *     byte frequency      chi^2 on the 256 values of each of the four bytes
 * Every test sums its statistic over its replications and turns the sum into ONE p-value (exact Poisson tails, chi^2).
 * Verdict per line, TestU01's convention: p outside [1e-3, 1 - 1e-3] "suspect", outside [1e-10, 1 - 1e-10] "FAIL" (a two-sided Poisson
 * p-value has only a lower end).
 *
 * Build / run:  gcc -O2 -fopenmp -o oracle/rng_battery oracle/rng_battery.c -lm && oracle/rng_battery [scale]
 *     scale 0 (default): 2^31.65 bytes per configuration (two minutes on 8 cores);  scale 3: 2^34.65 bytes per configuration, 2^37.65 in all;
 *     --control: a deliberately careless seeding, which the battery must reject (exit code 0 when it does). */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ generators */
#define M0 0xD2511F53u
#define M1 0xCD9E8D57u
#define W0 0x9E3779B9u
#define W1 0xBB67AE85u
#define MWC_A 4294584393ull
#define KEY1 0xCB435443u
#define CTR3 0x4D43475u
#define DRAWS 32

static void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1, int rounds) {
  for (int r = 0; r < rounds; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0; k1 += W1;
  }
}
/* generator 0: the product -- Philox4x32-7 seeds a multiply-with-carry lane generator */
static void stream_product(uint64_t id, uint32_t seed, uint32_t proj, uint32_t* out) {
  uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), proj, CTR3};
  philox4x32(c, seed, KEY1, 7);
  uint32_t x = c[0] ^ c[2];
  uint32_t cc = (uint32_t)(((uint64_t)(c[1] ^ c[3]) * (MWC_A - 1)) >> 32) + 1u;
  if (x == 0xFFFFFFFFu && cc == (uint32_t)(MWC_A - 1)) cc -= 1u;
  for (int k = 0; k < DRAWS; ++k) {
    const uint64_t t = MWC_A * x + cc;
    x = (uint32_t)t; cc = (uint32_t)(t >> 32);
    out[k] = x;
  }
}
/* generator 1: the yardstick -- Philox4x32-10 per draw, counter {id, projection, k / 4}, word k % 4 */
static void stream_philox10(uint64_t id, uint32_t seed, uint32_t proj, uint32_t* out) {
  for (int k4 = 0; k4 < DRAWS; k4 += 4) {
    uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), proj, (uint32_t)(k4 >> 2)};
    philox4x32(c, seed, KEY1, 10);
    for (int j = 0; j < 4; ++j) out[k4 + j] = c[j];
  }
}
/* generator 2: a NEGATIVE CONTROL (--control) -- the same multiply-with-carry step seeded straight from the history id, without the
 * Philox hash: neighbouring ids start from neighbouring states.  The battery must reject it (it does, in draw-major order, within
 * its smallest scale): evidence that the draw-major tests see what a careless per-history seeding gets wrong. */
static void stream_unhashed(uint64_t id, uint32_t seed, uint32_t proj, uint32_t* out) {
  uint32_t x = (uint32_t)id ^ (seed * 0x9E3779B9u) ^ (proj << 20), cc = (uint32_t)(id >> 32) + 1u;
  for (int k = 0; k < DRAWS; ++k) {
    const uint64_t t = MWC_A * x + cc;
    x = (uint32_t)t; cc = (uint32_t)(t >> 32);
    out[k] = x;
  }
}
typedef void (*stream_fn)(uint64_t, uint32_t, uint32_t, uint32_t*);

/* ------------------------------------------------------------------ serial source: ids x draws in one of the two orders */
#define BLOCK_LOG2 20
#define BLOCK_IDS (1u << BLOCK_LOG2)
typedef struct {
  stream_fn gen; int draw_major; uint32_t seed, proj;
  uint64_t next_id;         /* first id of the next refill */
  uint32_t* buf; size_t n, pos;  /* current window */
  uint64_t consumed;        /* words handed out */
} source_t;

static void source_init(source_t* s, stream_fn gen, int draw_major, uint32_t seed, uint32_t proj, uint64_t first_id) {
  memset(s, 0, sizeof *s);
  s->gen = gen; s->draw_major = draw_major; s->seed = seed; s->proj = proj; s->next_id = first_id;
  s->buf = (uint32_t*)malloc((size_t)(draw_major ? BLOCK_IDS : 1024u) * DRAWS * sizeof(uint32_t));
  if (!s->buf) { fprintf(stderr, "out of memory\n"); exit(2); }
}
static void source_free(source_t* s) { free(s->buf); s->buf = NULL; }
static void source_refill(source_t* s) {
  uint32_t tmp[DRAWS];
  if (s->draw_major) {
    for (uint32_t i = 0; i < BLOCK_IDS; ++i) {
      s->gen(s->next_id + i, s->seed, s->proj, tmp);
      for (int k = 0; k < DRAWS; ++k) s->buf[(size_t)k * BLOCK_IDS + i] = tmp[k];
    }
    s->next_id += BLOCK_IDS; s->n = (size_t)BLOCK_IDS * DRAWS;
  } else {
    for (uint32_t i = 0; i < 1024u; ++i) s->gen(s->next_id + i, s->seed, s->proj, s->buf + (size_t)i * DRAWS);
    s->next_id += 1024u; s->n = (size_t)1024u * DRAWS;
  }
  s->pos = 0;
}
static inline uint32_t source_next(source_t* s) {
  if (s->pos == s->n) source_refill(s);
  ++s->consumed;
  return s->buf[s->pos++];
}
/* ids a replication that draws `words` words must reserve (whole blocks) */
static uint64_t ids_for(uint64_t words, int draw_major) {
  const uint64_t per = draw_major ? (uint64_t)BLOCK_IDS : 1024u;
  const uint64_t ids = (words + DRAWS - 1) / DRAWS;
  return (ids + per - 1) / per * per;
}

/* ------------------------------------------------------------------ distributions */
static double gser(double a, double x) { double sum = 1.0 / a, del = sum, ap = a; for (int n = 0; n < 100000; ++n) { ap += 1.0; del *= x / ap; sum += del; if (fabs(del) < fabs(sum) * 1e-16) break; } return sum * exp(-x + a * log(x) - lgamma(a)); }
static double gcf(double a, double x) {
  double b = x + 1.0 - a, c = 1e300, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    const double an = -i * (i - a);
    b += 2.0; d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300; c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d; const double del = d * c; h *= del; if (fabs(del - 1.0) < 1e-16) break;
  }
  return exp(-x + a * log(x) - lgamma(a)) * h;
}
static double gammq(double a, double x) { if (x <= 0.0) return 1.0; return x < a + 1.0 ? 1.0 - gser(a, x) : gcf(a, x); }  /* upper regularised */
static double gammp(double a, double x) { if (x <= 0.0) return 0.0; return x < a + 1.0 ? gser(a, x) : 1.0 - gcf(a, x); }
static double chi2_upper(double x, double dof) { return gammq(0.5 * dof, 0.5 * x); }
/* Poisson(lambda): P(X <= x) = Q(x + 1, lambda), P(X >= x) = P(x, lambda).  Returned: the smaller tail doubled (two-sided, <= 1). */
static double poisson_two_sided(double x, double lambda) {
  const double lo = gammq(x + 1.0, lambda), hi = x > 0.0 ? gammp(x, lambda) : 1.0;
  const double p = 2.0 * (lo < hi ? lo : hi);
  return p > 1.0 ? 1.0 : p;
}
static int cmp_double(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }
static int cmp_u64(const void* a, const void* b) { const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return (x > y) - (x < y); }
static __attribute__((unused)) double ks_uniform(double* p, int n) {  /* two-sided Kolmogorov-Smirnov p-value of p[0..n) against U(0,1) */
  qsort(p, (size_t)n, sizeof(double), cmp_double);
  double d = 0.0;
  for (int i = 0; i < n; ++i) { const double a = (i + 1.0) / n - p[i], b = p[i] - (double)i / n; if (a > d) d = a; if (b > d) d = b; }
  const double t = (sqrt((double)n) + 0.12 + 0.11 / sqrt((double)n)) * d;
  double s = 0.0;
  for (int j = 1; j <= 100; ++j) s += 2.0 * ((j & 1) ? 1.0 : -1.0) * exp(-2.0 * j * j * t * t);
  return s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
}

/* ------------------------------------------------------------------ reporting */
static int n_suspect = 0, n_fail = 0, n_lines = 0;
/* one_sided: p is an upper-tail probability (chi^2): both ends are informative (near 1 = "too regular").  Otherwise p is the smaller
 * Poisson tail doubled: only small values speak against the generator (p = 1 is a statistic sitting on its mean). */
static const char* verdict(double p, int one_sided) {
  if (one_sided ? (p < 1e-10 || p > 1.0 - 1e-10) : (p < 2e-10)) { ++n_fail; return "FAIL"; }
  if (one_sided ? (p < 1e-3 || p > 1.0 - 1e-3) : (p < 2e-3)) { ++n_suspect; return "suspect"; }
  return "pass";
}
static void report(const char* test, const char* what, double stat, double p, int one_sided) {
  ++n_lines;
  printf("    %-18s %-62s statistic %12.6g   p = %.6f (%s)   %s\n", test, what, stat, p, one_sided ? "upper tail" : "two-sided", verdict(p, one_sided));
}

typedef struct { int shift, bits; const char* name; } bitsel_t;
static inline uint32_t take(uint32_t u, const bitsel_t* b) { return b->bits == 32 ? u : ((u >> b->shift) & ((1u << b->bits) - 1u)); }

typedef struct { stream_fn gen; int draw_major; uint32_t seed, proj; uint64_t cursor; uint64_t consumed; } config_t;
/* A test's replications are dealt to LANES serial sources (a fixed number: the result does not depend on the thread count); a
 * lane's replications follow each other in ONE sequence, so that in draw-major order they reach every depth of the streams.
 * reserve(): ids for LANES lanes of `words_per_lane` words each; returns the first id of lane 0 and the stride between lanes. */
#define LANES 16
static uint64_t reserve(config_t* c, uint64_t words_per_lane, uint64_t* stride) {
  *stride = ids_for(words_per_lane, c->draw_major);
  const uint64_t first = c->cursor;
  c->cursor += *stride * (uint64_t)LANES;
  c->consumed += words_per_lane * (uint64_t)LANES;
  return first;
}

/* ------------------------------------------------------------------ birthday spacings */
static void test_birthday(config_t* c, const bitsel_t* b, int t, int log2n, int reps) {
  const int cell_bits = t * b->bits;
  const size_t n = (size_t)1 << log2n;
  const double lambda = pow(2.0, 3.0 * log2n - 2.0 - cell_bits);
  const int per_lane = reps / LANES;
  uint64_t stride; const uint64_t first = reserve(c, (uint64_t)n * t * per_lane, &stride);
  double total = 0.0;
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
  for (int lane = 0; lane < LANES; ++lane) {
    source_t s; source_init(&s, c->gen, c->draw_major, c->seed, c->proj, first + stride * (uint64_t)lane);
    uint64_t* cell = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (int r = 0; r < per_lane; ++r) {
      for (size_t i = 0; i < n; ++i) { uint64_t v = 0; for (int j = 0; j < t; ++j) v = (v << b->bits) | take(source_next(&s), b); cell[i] = v; }
      qsort(cell, n, sizeof(uint64_t), cmp_u64);
      for (size_t i = n - 1; i > 0; --i) cell[i] -= cell[i - 1];   /* the n - 1 spacings between neighbours, in cell[1..n) */
      qsort(cell + 1, n - 1, sizeof(uint64_t), cmp_u64);
      double y = 0.0; for (size_t i = 2; i < n; ++i) y += (cell[i] == cell[i - 1]);   /* spacings minus distinct spacings */
      total += y;
    }
    free(cell); source_free(&s);
  }
  char what[96]; snprintf(what, sizeof what, "%s, t=%d, n=2^%d, k=2^%d, %d x lambda %.3g", b->name, t, log2n, cell_bits, per_lane * LANES, lambda);
  report("birthday spacings", what, total, poisson_two_sided(total, lambda * per_lane * LANES), 0);
}

/* ------------------------------------------------------------------ collisions (Knuth 3.3.2 I) */
static void test_collision(config_t* c, const bitsel_t* b, int log2n, int reps) {
  const int per = b->bits >= 20 ? 20 : b->bits, t = 40 / per;   /* 40-bit urn number from the TOP `per` bits of t consecutive values */
  const size_t n = (size_t)1 << log2n;
  const double k = pow(2.0, per * t);
  const double mean = (double)n - k * (-expm1((double)n * log1p(-1.0 / k)));   /* n - k (1 - (1 - 1/k)^n) */
  const int per_lane = reps / LANES;
  uint64_t stride; const uint64_t first = reserve(c, (uint64_t)n * t * per_lane, &stride);
  double total = 0.0;
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
  for (int lane = 0; lane < LANES; ++lane) {
    source_t s; source_init(&s, c->gen, c->draw_major, c->seed, c->proj, first + stride * (uint64_t)lane);
    uint64_t* urn = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (int r = 0; r < per_lane; ++r) {
      for (size_t i = 0; i < n; ++i) { uint64_t v = 0; for (int j = 0; j < t; ++j) v = (v << per) | (take(source_next(&s), b) >> (b->bits - per)); urn[i] = v; }
      qsort(urn, n, sizeof(uint64_t), cmp_u64);
      double col = 0.0; for (size_t i = 1; i < n; ++i) col += (urn[i] == urn[i - 1]);
      total += col;
    }
    free(urn); source_free(&s);
  }
  char what[96]; snprintf(what, sizeof what, "%s, top %d bits x %d, n=2^%d, k=2^%d, %d x mean %.2f", b->name, per, t, log2n, per * t, per_lane * LANES, mean);
  report("collisions", what, total, poisson_two_sided(total, mean * per_lane * LANES), 0);
}

/* ------------------------------------------------------------------ matrix rank over GF(2) */
static int rank_gf2(uint32_t* row, int m) {
  int rank = 0;
  for (int col = m - 1; col >= 0 && rank < m; --col) {
    int piv = -1;
    for (int i = rank; i < m; ++i) if ((row[i] >> col) & 1u) { piv = i; break; }
    if (piv < 0) continue;
    const uint32_t tmp = row[piv]; row[piv] = row[rank]; row[rank] = tmp;
    for (int i = 0; i < m; ++i) if (i != rank && ((row[i] >> col) & 1u)) row[i] ^= row[rank];
    ++rank;
  }
  return rank;
}
static double rank_prob(int m, int r) {  /* P(rank = r) of a random m x m matrix over GF(2) */
  double lg = (double)(r * (2 * m - r) - m * m) * log(2.0);
  for (int i = 0; i < r; ++i) lg += 2.0 * log1p(-pow(2.0, i - m)) - log1p(-pow(2.0, i - r));
  return exp(lg);
}
static void test_rank(config_t* c, const bitsel_t* b, long matrices) {
  const int m = b->bits;
  uint64_t stride; const int reps = LANES; const long per = matrices / reps;
  const uint64_t first = reserve(c, (uint64_t)per * m, &stride);
  double cnt[3] = {0, 0, 0};
#pragma omp parallel for schedule(dynamic)
  for (int r = 0; r < reps; ++r) {
    source_t s; source_init(&s, c->gen, c->draw_major, c->seed, c->proj, first + stride * (uint64_t)r);
    double local[3] = {0, 0, 0}; uint32_t row[32];
    for (long q = 0; q < per; ++q) {
      for (int i = 0; i < m; ++i) row[i] = take(source_next(&s), b);
      const int rk = rank_gf2(row, m);
      local[rk == m ? 2 : (rk == m - 1 ? 1 : 0)] += 1.0;
    }
#pragma omp critical
    { cnt[0] += local[0]; cnt[1] += local[1]; cnt[2] += local[2]; }
    source_free(&s);
  }
  const double N = (double)per * reps, p2 = rank_prob(m, m), p1 = rank_prob(m, m - 1), p0 = 1.0 - p1 - p2;
  const double e[3] = {N * p0, N * p1, N * p2};
  double x2 = 0.0; for (int i = 0; i < 3; ++i) x2 += (cnt[i] - e[i]) * (cnt[i] - e[i]) / e[i];
  char what[96]; snprintf(what, sizeof what, "%s, %d x %d matrices, N = %.0f", b->name, m, m, N);
  report("matrix rank", what, x2, chi2_upper(x2, 2.0), 1);
}

/* ------------------------------------------------------------------ linear complexity (NIST SP 800-22 2.10) */
static int berlekamp_massey(const unsigned char* s, int n) {
  unsigned char cpoly[1024], bpoly[1024], tpoly[1024];
  memset(cpoly, 0, sizeof cpoly); memset(bpoly, 0, sizeof bpoly); cpoly[0] = bpoly[0] = 1;
  int L = 0, m = -1;
  for (int i = 0; i < n; ++i) {
    int d = s[i];
    for (int j = 1; j <= L; ++j) d ^= cpoly[j] & s[i - j];
    if (d) {
      memcpy(tpoly, cpoly, (size_t)n + 1);
      for (int j = 0; j + i - m <= n; ++j) cpoly[j + i - m] ^= bpoly[j];
      if (2 * L <= i) { L = i + 1 - L; m = i; memcpy(bpoly, tpoly, (size_t)n + 1); }
    }
  }
  return L;
}
static void test_linear_complexity(config_t* c, int bit, long blocks) {
  const int M = 1000, reps = LANES; const long per = blocks / reps;
  static const double pi[7] = {0.010417, 0.03125, 0.125, 0.5, 0.25, 0.0625, 0.020833};
  uint64_t stride; const uint64_t first = reserve(c, (uint64_t)per * M, &stride);
  double cnt[7] = {0, 0, 0, 0, 0, 0, 0};
  const double mu = M / 2.0 + (9.0 + ((M + 1) % 2 ? -1.0 : 1.0)) / 36.0 - (M / 3.0 + 2.0 / 9.0) / pow(2.0, M);
#pragma omp parallel for schedule(dynamic)
  for (int r = 0; r < reps; ++r) {
    source_t s; source_init(&s, c->gen, c->draw_major, c->seed, c->proj, first + stride * (uint64_t)r);
    double local[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned char seq[1000];
    for (long q = 0; q < per; ++q) {
      for (int i = 0; i < M; ++i) seq[i] = (unsigned char)((source_next(&s) >> bit) & 1u);
      const int L = berlekamp_massey(seq, M);
      const double T = ((M % 2) ? -1.0 : 1.0) * (L - mu) + 2.0 / 9.0;
      const int k = T <= -2.5 ? 0 : T <= -1.5 ? 1 : T <= -0.5 ? 2 : T <= 0.5 ? 3 : T <= 1.5 ? 4 : T <= 2.5 ? 5 : 6;
      local[k] += 1.0;
    }
#pragma omp critical
    for (int k = 0; k < 7; ++k) cnt[k] += local[k];
    source_free(&s);
  }
  const double N = (double)per * reps;
  double x2 = 0.0; for (int k = 0; k < 7; ++k) x2 += (cnt[k] - N * pi[k]) * (cnt[k] - N * pi[k]) / (N * pi[k]);
  char what[96]; snprintf(what, sizeof what, "bit %d of each output, %.0f blocks of %d bits", bit, N, M);
  report("linear complexity", what, x2, chi2_upper(x2, 6.0), 1);
}

/* ------------------------------------------------------------------ byte frequencies */
static void test_bytes(config_t* c, long words) {
  const int reps = LANES; const long per = words / reps;
  uint64_t stride; const uint64_t first = reserve(c, (uint64_t)per, &stride);
  static double cnt[4][256]; memset(cnt, 0, sizeof cnt);
#pragma omp parallel for schedule(dynamic)
  for (int r = 0; r < reps; ++r) {
    source_t s; source_init(&s, c->gen, c->draw_major, c->seed, c->proj, first + stride * (uint64_t)r);
    double local[4][256]; memset(local, 0, sizeof local);
    for (long q = 0; q < per; ++q) { const uint32_t u = source_next(&s); for (int j = 0; j < 4; ++j) local[j][(u >> (8 * j)) & 255u] += 1.0; }
#pragma omp critical
    for (int j = 0; j < 4; ++j) for (int v = 0; v < 256; ++v) cnt[j][v] += local[j][v];
    source_free(&s);
  }
  const double N = (double)per * reps, e = N / 256.0;
  for (int j = 0; j < 4; ++j) {
    double x2 = 0.0; for (int v = 0; v < 256; ++v) x2 += (cnt[j][v] - e) * (cnt[j][v] - e) / e;
    char what[96]; snprintf(what, sizeof what, "byte %d (bits %d..%d), N = %.0f", j, 8 * j, 8 * j + 7, N);
    report("byte frequency", what, x2, chi2_upper(x2, 255.0), 1);
  }
}

/* known answers: the first words of a few streams, for tests/test_fast_rng.py to compare with oracle/fast_rng.py */
static void print_known_answers(void) {
  static const uint64_t ids[4] = {0ull, 1ull, 0xFFFFFFFFull, (1ull << 40) + 12345ull};
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < 4; ++i) {
      uint32_t out[DRAWS]; (g ? stream_philox10 : stream_product)(ids[i], 42u, 893u, out);
      printf("kat generator %d id %llu seed 42 projection 893:", g, (unsigned long long)ids[i]);
      for (int k = 0; k < 6; ++k) printf(" %08x", out[k]);
      printf("\n");
    }
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--kat")) { print_known_answers(); return 0; }
  const int control = argc > 1 && !strcmp(argv[1], "--control");
  const int scale = (argc > 1 && !control) ? atoi(argv[1]) : 0;   /* every replication count doubles per unit */
  const int f = 1 << (scale < 0 ? 0 : (scale > 4 ? 4 : scale));
  static const bitsel_t all32 = {0, 32, "32 bits"}, up24 = {8, 24, "upper 24 bits"}, low8 = {0, 8, "low 8 bits"};
  printf("# rng_battery scale %d, %d threads; seed 42; blocks of 2^%d ids in draw-major order; %d draws per history\n", scale, omp_get_max_threads(), BLOCK_LOG2, DRAWS);
  uint64_t grand = 0;
  for (int g = control ? 2 : 0; g < (control ? 3 : 2); ++g)
    for (int order = 0; order < 2; ++order)
      for (int pj = 0; pj < (control ? 1 : 2); ++pj) {
        config_t c = {g == 2 ? stream_unhashed : (g ? stream_philox10 : stream_product), order, 42u, pj ? 893u : 0u, 0ull, 0ull};
        const int before_s = n_suspect, before_f = n_fail;
        printf("%s | %s | projection %u\n", g == 2 ? "NEGATIVE CONTROL: multiply-with-carry seeded from the id without the Philox hash" : g ? "Philox4x32-10 per draw (yardstick)" : "PRODUCT: Philox4x32-7 seeding + multiply-with-carry", order ? "draw-major" : "history-major", c.proj);
        test_birthday(&c, &all32, 2, 22, 16 * f);
        test_birthday(&c, &up24, 2, 17, 128 * f);
        test_birthday(&c, &low8, 4, 12, 512 * f);
        test_collision(&c, &all32, 22, 16 * f);
        test_collision(&c, &up24, 22, 16 * f);
        test_collision(&c, &low8, 22, 16 * f);
        test_rank(&c, &all32, 262144L * f);
        test_rank(&c, &up24, 262144L * f);
        test_rank(&c, &low8, 1048576L * f);
        test_linear_complexity(&c, 0, 8192L * f);
        test_linear_complexity(&c, 8, 8192L * f);
        test_linear_complexity(&c, 31, 8192L * f);
        test_bytes(&c, 16777216L * f);
        printf("    -> %.3f GiB of outputs (2^%.2f bytes), history ids 0 .. %llu; suspect %d, FAIL %d\n", c.consumed * 4.0 / 1073741824.0, log2(c.consumed * 4.0),
               (unsigned long long)c.cursor, n_suspect - before_s, n_fail - before_f);
        grand += c.consumed;
        fflush(stdout);
      }
  printf("# total %.2f GiB (2^%.2f bytes) in %d statistics: %d suspect (upper-tail p outside [1e-3, 1 - 1e-3] or two-sided p < 2e-3; expected by chance: %.1f), %d FAIL (beyond 1e-10)\n",
         grand * 4.0 / 1073741824.0, log2(grand * 4.0), n_lines, n_suspect, n_lines * 2e-3, n_fail);
  if (control) return n_fail ? 0 : 1;   /* the control run succeeds when the battery rejects the generator */
  return n_fail ? 1 : 0;
}
