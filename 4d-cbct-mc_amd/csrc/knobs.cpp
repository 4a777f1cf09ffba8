// knobs.cpp -- the ONE place the engine reads its environment.  Every MCGPU_* variable the library, the executable or the Python
// mirror understands is a row of kKnobs (name, type, default, what it selects); code asks by name through knob_*() -- a name that
// is not registered is a programming error and throws -- and variables of the environment that start with MCGPU_ and are NOT
// registered (a misspelt knob) produce one warning line per process instead of being ignored silently.
// `MC-GPU_v1.3.x --knobs` and mcgpu_knob_table() print the table; INTEGRATION.md carries it; the PMC stamp of the bench
// (bench_legs/common.py: knob_environment) is built from the same list.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "host_model.hpp"
#include "knobs.hpp"
#include "../../include/mcgpu_amd.h"

extern char** environ;

namespace mcgpu {

// type: i = integer, f = seconds (float), b = switch (set to anything = on), s = string
// scope: K = picks a kernel variant or its schedule (part of the PMC stamp), H = host-side pipeline, T = test hook, P = read by Python
const KnobSpec kKnobs[] = {
    {"MCGPU_AMD_LIB", 's', 'P', "4d-cbct-mc_amd/libmcgpu_amd.so", "library the Python mirror loads (engine.py: load_library); A/B builds and the diagnostic library go through it"},
    {"MCGPU_FAST_SCHED", 'i', 'K', "0", "FAST scheduler: 0 per-wave pools, 1 workgroup-level pool (bit-identical, 14-23 % slower; fixed when the model is uploaded)"},
    {"MCGPU_SEGMENT_LOOP", 'i', 'K', "-1", "flight segment as an inner loop: 1 on, 0 off, -1 host heuristic (tile records, or >= 20 electron shells per material on average)"},
    {"MCGPU_TILE_RECORDS", 'i', 'K', "auto", "16-byte tile records (two to four palette entries + a 64-bit mask per 4x4x4 tile) as second level of the u8 volume: 1 on, 0 off; default on when the tiles of the mixed bricks exceed 8 MiB (thorax +9 %, CT-textured +29 %); identical tallies"},
    {"MCGPU_SUB_BRICKS", 'i', 'K', "0", "4-bit sub-brick codes as second level (superseded by the tile records; kept for A/B)"},
    {"MCGPU_MAX_BRICKS", 'i', 'K', "fit", "upper limit of the LDS brick grid (a coarser grid frees LDS); default: what is left of 80 KB after the tables"},
    {"MCGPU_NO_BRACKETS", 'b', 'K', "off", "no cross-section brackets in LDS: every material change asks the exact table"},
    {"MCGPU_NO_EXTERIOR", 'b', 'K', "off", "no analytic exterior hop / source entry: delta-tracking everywhere, as the reference"},
    {"MCGPU_EXTERIOR_MODE", 'i', 'K', "3", "bit 0: exterior hop of photons in flight, bit 1: analytic source entry"},
    {"MCGPU_NO_ELLIPSE", 'b', 'K', "off", "object region = bounding box only (no elliptic cylinder; default: cylinder when >= 5 % of the bricks lie outside it)"},
    {"MCGPU_THRESH_COMPTON", 'i', 'K', "40", "lanes a Compton batch waits for (1..64); setting any schedule knob also switches the scan's autotune off"},
    {"MCGPU_THRESH_RAYLEIGH", 'i', 'K', "12", "lanes a Rayleigh batch waits for"},
    {"MCGPU_THRESH_NEW", 'i', 'K', "44", "lanes a tally + source batch waits for"},
    {"MCGPU_FLYABLE_LOW", 'i', 'K', "12", "below this many flyable histories the wave drains its pending kinds"},
    {"MCGPU_SWAP_BATCH", 'i', 'K', "40", "lanes that may park before the next scheduling point"},
    {"MCGPU_HOLD_Q", 'i', 'K', "6", "a flight segment ends when all but hold_q/16 of its starting lanes have parked (0..15)"},
    {"MCGPU_SLOT_TRADE", 'i', 'K', "3", "bit 0: trade slots for flyable histories, bit 1: for the service batches"},
    {"MCGPU_NO_AUTOTUNE", 'b', 'K', "off", "scan driver: keep the default schedule instead of probing the three presets on 6e6 histories"},
    {"MCGPU_BLOCKS_PER_CU", 'i', 'K', "occupancy", "resident FAST workgroups per CU (default: what the occupancy query reports, normally 2)"},
    {"MCGPU_GRID_SPARE_PERCENT", 'i', 'K', "0", "extra workgroups beyond the resident grid, in percent"},
    {"MCGPU_COMPAT_THRESH_COMPTON", 'i', 'K', "-1", "COMPAT kernel: lanes a Compton batch waits for (-1: built-in 32, or 48 when the materials average >= 20 electron shells)"},
    {"MCGPU_COMPAT_THRESH_RAYLEIGH", 'i', 'K', "-1", "COMPAT kernel: Rayleigh batch threshold (built-in 4 / 8)"},
    {"MCGPU_COMPAT_THRESH_NEW", 'i', 'K', "-1", "COMPAT kernel: tally + source batch threshold (built-in 24 / 16)"},
    {"MCGPU_COMPAT_THRESH_TAKE", 'i', 'K', "-1", "COMPAT kernel: lanes whose parked history could fly while their register history cannot, to exchange the two (built-in 2)"},
    {"MCGPU_COMPAT_STATS", 'i', 'K', "0", "COMPAT kernel section counters (diagnostic build -DMC_COMPAT_STATS only)"},
    {"MCGPU_IGNORE_VOXBIN", 'b', 'H', "off", "parse geometry.vox(.gz) even when a newer geometry.voxbin sidecar lies beside it"},
    {"MCGPU_ASCII_HOST", 'b', 'H', "off", "format the reference's ASCII projection files on the host threads instead of the device (A/B)"},
    {"MCGPU_ASCII_WRITERS", 'i', 'H', "slots", "ASCII files in flight (1..MCGPU_ASCII_SLOTS)"},
    {"MCGPU_PINNED_COHERENT", 'b', 'H', "off", "coherent pinned host buffers for the downloads (default: non-coherent)"},
    {"MCGPU_FDK_DIRECT_RAMP", 'b', 'H', "off", "FDK ramp filter as a direct LDS convolution instead of hipFFT"},
    {"MCGPU_REDUCE", 's', 'H', "auto", "multi-device tally sum: `rccl` = one ncclReduce per projection; default: tally exchange, else projection sharding"},
    {"MCGPU_RCCL_LIBRARY", 's', 'H', "librccl.so.1", "RCCL library the reduction route opens with dlopen"},
    {"MCGPU_EXCHANGE_POLICY", 'i', 'H', "1", "tally exchange: 1 the owner of a projection rotates over the ranks, 0 rank 0 owns every projection"},
    {"MCGPU_EXCHANGE_TIMEOUT_S", 'f', 'H', "120", "seconds a rank waits for a peer's counter before the exchange gives up"},
    {"MCGPU_EXCHANGE_FAIL_PROBE", 'b', 'T', "off", "test hook: the exchange's set-up probe reports failure (exercises the fallback chain)"},
    {"MCGPU_RCCL_FAIL", 'b', 'T', "off", "test hook: the RCCL route's set-up reports failure"},
    {"MCGPU_RNG_TEST_LOG2", 'i', 'P', "20", "tests/test_fast_rng.py: log2 of the history ids of the statistical test"},
    {"MCGPU_TEST_CACHE", 's', 'P', "/tmp/mcgpu_amd_test_cache", "tests/cases.py: directory of the generated test inputs"},
};
const int kNumKnobs = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));

static const char* env_value(const char* name) { return getenv(name); }  // the engine's only read of the environment
static const KnobSpec* find_knob(const char* name) {
  for (int i = 0; i < kNumKnobs; ++i)
    if (!strcmp(kKnobs[i].name, name)) return &kKnobs[i];
  return nullptr;
}

const char* knob_str(const char* name) {
  if (!find_knob(name)) throw Error(-9, std::string("!!ERROR!! internal: environment knob ") + name + " is not in the registry (csrc/knobs.cpp)");
  return env_value(name);
}
bool knob_set(const char* name) { return knob_str(name) != nullptr; }
int knob_int(const char* name, int dflt) { const char* v = knob_str(name); return v ? atoi(v) : dflt; }
double knob_float(const char* name, double dflt) { const char* v = knob_str(name); return v ? atof(v) : dflt; }

// One line per unregistered MCGPU_* variable, once per process, on stdout like the rest of the engine's log -- worded without the
// substring the reference's caller takes for a failed run (cbctmc/mc/simulation.py:204 greps for "error", any case).
void knobs_warn_unknown() {
  static std::once_flag once;
  std::call_once(once, [] {
    for (char** e = environ; e && *e; ++e) {
      if (strncmp(*e, "MCGPU_", 6) != 0) continue;
      const char* eq = strchr(*e, '=');
      const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
      if (find_knob(name.c_str())) continue;
      printf("       [warning] environment variable %s is not a knob of this engine and is ignored (`MC-GPU_v1.3.x --knobs` lists them)\n", name.c_str());
      fflush(stdout);
    }
  });
}

}  // namespace mcgpu

using namespace mcgpu;

// The table as text (one knob per line: name, type, scope, default, current value, description), NUL-terminated into buf;
// returns the number of bytes the whole table needs (call with cap = 0 to size the buffer).
size_t mcgpu_knob_table(char* buf, size_t cap) {
  std::string t;
  for (int i = 0; i < kNumKnobs; ++i) {
    const KnobSpec& k = kKnobs[i];
    const char* cur = env_value(k.name);
    t += k.name; t += '\t'; t += k.type; t += '\t'; t += k.scope; t += '\t'; t += k.dflt; t += '\t'; t += cur ? cur : ""; t += '\t'; t += k.what; t += '\n';
  }
  if (buf && cap > 0) {
    const size_t n = t.size() < cap - 1 ? t.size() : cap - 1;
    memcpy(buf, t.data(), n);
    buf[n] = 0;
  }
  return t.size() + 1;
}
