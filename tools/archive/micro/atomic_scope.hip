// Round 6 experiment: can the detector tally's scattered 64-bit adds run in the XCD-local L2 instead of at the memory side?
// The FAST kernel issues one agent-scope global_atomic_add_x2 per detected photon; on a multi-XCD part those are executed at the memory
// side (2.37e10/s chip-wide: the wall the Catphan launch sits at 0.87 of).  A workgroup-scope add may be executed in the issuing XCD's
// L2 -- usable if every XCD tallies into a replica of its own (8 x 45 MB) that a pass at the end of the projection sums.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/micro/atomic_scope.hip -o build/atomic_scope && build/atomic_scope
// Prints adds per second for {agent scope, one tally} and {workgroup scope, one replica per XCD}, uniform over the tally and with the
// Catphan's concentration (60 % of the adds into the 6.3 MB of the primary image), and checks the replicas' sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr unsigned int kWords = 4u * 1848u * 768u;   // uint64 words of one tally
constexpr unsigned int kPrimary = 768u * 1024u;      // words of the primary image's illuminated part

template <int SCOPE, bool CONCENTRATED>
__global__ void scatter(unsigned long long* img, int per_thread, unsigned int seed, unsigned int* xcc_seen) {
  const unsigned int xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u;   // XCC_ID, bits 0..3
  unsigned long long* tally = SCOPE == 0 ? img : img + (size_t)xcc * kWords;
  if (threadIdx.x == 0) xcc_seen[xcc] = 1u;
  unsigned int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  for (int i = 0; i < per_thread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    unsigned int w = x % kWords;
    if (CONCENTRATED && (x >> 24) < 154u) w = (x >> 4) % kPrimary;   // 60 % into the primary image
    if (SCOPE == 0) __hip_atomic_fetch_add(tally + w, (unsigned long long)(x & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(tally + w, (unsigned long long)(x & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void sum_replicas(const unsigned long long* img, int replicas, unsigned long long* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kWords) return;
  unsigned long long s = 0;
  for (int r = 0; r < replicas; ++r) s += img[(size_t)r * kWords + i];
  out[i] = s;
}
__global__ void checksum(const unsigned long long* a, unsigned long long* total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kWords && a[i]) atomicAdd(total, a[i]);
}

template <int SCOPE, bool CONC>
double run(unsigned long long* img, unsigned long long* summed, unsigned long long* total, unsigned int* seen, const char* what) {
  const int blocks = 256 * 8, threads = 256, per = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f, t_sum = 0.f;
  unsigned long long tot = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(img, 0, (size_t)kWords * 8 * 8); hipMemset(total, 0, 8); hipMemset(seen, 0, 32);
    hipDeviceSynchronize();
    hipEventRecord(a);
    scatter<SCOPE, CONC><<<blocks, threads>>>(img, per, 17u, seen);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
    hipEventRecord(a);
    sum_replicas<<<(kWords + 255) / 256, 256>>>(img, SCOPE == 0 ? 1 : 8, summed);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&t_sum, a, b);
    checksum<<<(kWords + 255) / 256, 256>>>(summed, total);
    hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost);
  }
  unsigned int s[8]; hipMemcpy(s, seen, 32, hipMemcpyDeviceToHost);
  int nx = 0; for (int k = 0; k < 8; ++k) nx += s[k] ? 1 : 0;
  const double rate = (double)blocks * threads * per / (best * 1e-3);
  printf("%-58s %8.3f ms  %.3e adds/s   sum of the tally %llu   replicas summed in %.3f ms   XCDs seen %d\n", what, best, rate, tot, t_sum, nx);
  return rate;
}

int main() {
  unsigned long long *img, *summed, *total; unsigned int* seen;
  hipMalloc(&img, (size_t)kWords * 8 * 8); hipMalloc(&summed, (size_t)kWords * 8); hipMalloc(&total, 8); hipMalloc(&seen, 32);
  run<0, false>(img, summed, total, seen, "agent scope, one tally, uniform");
  run<1, false>(img, summed, total, seen, "workgroup scope, replica per XCD, uniform");
  run<0, true>(img, summed, total, seen, "agent scope, one tally, 60 % into the primary image");
  run<1, true>(img, summed, total, seen, "workgroup scope, replica per XCD, 60 % into the primary image");
  return 0;
}
