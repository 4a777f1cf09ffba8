// exchange.cpp -- the tally exchange between the GPUs of one node: the ONE N > 1 data path of the engine, used by the
// drop-in executable (scan.cpp: mcgpu_run_scan_multi, contexts of one process) and by bench.py (one process per GPU).
//
// What it replaces: the reference copies every rank's 45 MB detector tally to the host and sums them with one blocking
// MPI_Reduce per projection on rank 0 (docker/mcgpu/MC-GPU_v1.3.cu:1006-1024).  What it does instead, shaped by three
// measurements on MI355X:
//   * a kernel that is resident while the persistent tracking grid is dispatched costs that launch up to 30 %, and a plain
//     device-to-device hipMemcpyAsync IS a kernel (45 MB beside a busy grid: 11 ms instead of 0.02 ms);
//   * a copy ENGINE transfer (hipMemcpyDeviceToDeviceNoCU) moves the same 45 MB in 0.75 ms at 60 GB/s beside the busy grid
//     and leaves the grid's time unchanged (profiles/r03a_ipc_probe_two_processes_one_gpu.txt);
//   * IPC memory handles and interprocess events work between processes on this pool, including hipStreamWaitEvent on an
//     opened event.
// So every projection ("step") has an OWNER rank.  A rank that does not own step i pushes its tally of i into the owner's
// landing buffer with a copy engine while its next projection is being tracked; the owner adds the landed tallies to its
// own with one fused pass, ordered on its tracking stream behind the kernel of step i + 1.  Nothing but that one pass
// (0.4 GB of traffic at N = 8, about 0.1 ms) is ever exposed; xGMI is point-to-point, so N - 1 pushes into one owner use
// N - 1 different links.  Owner policy: 0 = always rank 0 (the reference's root), 1 = step mod world (every rank owns 1/N of
// the projections: the exposed pass shrinks to 1/N per step and every link of the node carries the same load).
//
// Ordering.  Device side: events.  pushed[r->o][q] is recorded on r's copy stream behind a push, consumed[o][q] on o's
// tracking stream behind the pass that read landing parity q; both are interprocess events when the peer lives in another
// process.  A stream wait on an event means "the latest record the runtime knew of when the wait was ISSUED", so the host
// side must make sure the record has been issued first: two monotonic counters per rank in a small host region every rank
// maps (`shared`: plain memory for contexts of one process, a /dev/shm file for processes) -- push_issued[r][o] and
// collect_issued[o].  A wait spins on the counter (bounded: an error after 120 s, never a hang), then issues the stream
// wait.  Because an owner cannot collect index k before every peer issued push k, and a peer cannot push k + 2 before the
// owner issued collect k, the "latest record" is always the intended one.
//
// Units, tallies and sums are integers: the reduced tally equals the single-GPU tally bit for bit, for any world size
// (tests/test_exchange.py, bench.py `check.sharded_equals_single`).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcgpu_amd.h"
#include "knobs.hpp"

namespace mcgpu {
hipError_t launch_accumulate_many(unsigned long long* dst, const unsigned long long* const* src, int n, size_t words, hipStream_t stream);  // finalize.hip
}

extern "C" void mcgpu_set_last_error_(const char* message);  // engine.cpp

namespace {

struct XError {
  int code;
  std::string msg;
};
#define X_HIP(expr)                                                                                              \
  do {                                                                                                           \
    hipError_t _e = (expr);                                                                                      \
    if (_e != hipSuccess) throw XError{-1, std::string("!!HIP ERROR!! ") + #expr + ": " + hipGetErrorString(_e)}; \
  } while (0)
// A HIP call of the exchange's SET-UP between two ranks, labelled with the mechanism it belongs to, so that the first log of a real
// node says which of the three things the exchange needs is missing there: the peer's landing memory mapped through an IPC memory
// handle, stream waits on the peer's interprocess events, or a copy-engine transfer into that memory.
#define X_STAGE(stage, rank, peer, expr)                                                                                         \
  do {                                                                                                                           \
    hipError_t _e = (expr);                                                                                                      \
    if (_e != hipSuccess)                                                                                                        \
      throw XError{-1, std::string("!!HIP ERROR!! tally exchange set-up, stage [") + (stage) + "] rank " + std::to_string(rank) + " -> peer " + \
                           std::to_string(peer) + ": " + #expr + ": " + hipGetErrorString(_e)};                                  \
  } while (0)
#define X_REQUIRE(cond, msg)                       \
  do {                                             \
    if (!(cond)) throw XError{-1, std::string(msg)}; \
  } while (0)

constexpr int kMaxWorld = 64;
constexpr int kHandle = 64;  // sizeof(hipIpcMemHandle_t) == sizeof(hipIpcEventHandle_t) == 64
static_assert(sizeof(hipIpcMemHandle_t) == kHandle && sizeof(hipIpcEventHandle_t) == kHandle, "IPC handle size");

// host counters of one rank in the shared region (one cache line per counter block start; world <= 64)
struct alignas(64) Mailbox {
  std::atomic<long long> collect_issued;            // collects this rank has issued as an owner
  std::atomic<long long> push_issued[kMaxWorld];    // pushes this rank has issued to owner o
  std::atomic<long long> closing;                   // this rank is being destroyed / has failed: peers stop waiting
};

}  // namespace

struct mcgpu_exchange {
  int device = -1, rank = 0, world = 1, policy = 0;
  bool local = false;  // every rank lives in this process: plain events instead of interprocess ones
  size_t words = 0;
  Mailbox* boxes = nullptr;  // [world], in the shared region
  hipStream_t copy = nullptr;    // pushes (copy engine)
  // An interprocess event is waited for and recorded through host callbacks of the runtime: cheap, but not something to put
  // on the tracking stream.  All of them live on this side stream; the tracking stream only ever sees local events.
  hipStream_t gather = nullptr;
  hipEvent_t landed[2] = {nullptr, nullptr};    // every peer's push into landing parity q has arrived (recorded on `gather`)
  hipEvent_t added[2] = {nullptr, nullptr};     // the fused add over landing parity q has run (recorded on the tracking stream)
  unsigned long long* tally[2] = {nullptr, nullptr};
  unsigned long long* landing = nullptr;           // [world][2][words] on this device (slot of peer r, parity q); own row unused
  unsigned long long* peer_landing[kMaxWorld] = {};  // mapping of peer o's landing block in this process
  bool peer_mapped_ipc[kMaxWorld] = {};
  hipEvent_t tracked[2] = {nullptr, nullptr};      // tracking launch of the step with that tally parity has been enqueued
  hipEvent_t pushed_local[2] = {nullptr, nullptr}; // the push that reads tally[b] is done (local event)
  bool pushed_local_valid[2] = {false, false};
  hipEvent_t pushed[kMaxWorld][2] = {};            // mine: push to owner o, landing parity q (interprocess)
  hipEvent_t consumed[2] = {nullptr, nullptr};     // mine as an owner: landing parity q has been added (interprocess)
  hipEvent_t peer_pushed[kMaxWorld][2] = {};       // peer r's pushed[r -> me][q]
  hipEvent_t peer_consumed[kMaxWorld][2] = {};     // owner o's consumed[q]
  bool connected[kMaxWorld] = {};
  bool opened_events[kMaxWorld] = {};
  const unsigned long long** src_table[2] = {nullptr, nullptr};  // device arrays of the landing pointers of parity q for the fused add
  hipEvent_t t_push0 = nullptr, t_push1 = nullptr, t_acc0 = nullptr, t_acc1 = nullptr;  // timing of the last push / pass
  bool timed_push = false, timed_acc = false;
  long long pushes = 0, collects = 0;
  long long last_submitted = -1, last_collected = -1;  // steps come in increasing order, each once
  double wait_seconds = 0.0;  // host time spent spinning on counters

  int owner(long long step) const { return policy == 1 ? (int)(step % world) : 0; }
  // index of `step` among the steps its owner owns
  long long owned_index(long long step) const { return policy == 1 ? step / world : step; }
  unsigned long long* slot(unsigned long long* base, int from, int q) const { return base + ((size_t)from * 2 + (size_t)q) * words; }
};

namespace {

// how long a rank waits for a peer's counter before it gives up with an error (MCGPU_EXCHANGE_TIMEOUT_S, default 120)
double wait_limit_seconds() {
  static const double limit = [] {
    const double s = mcgpu::knob_float("MCGPU_EXCHANGE_TIMEOUT_S", 120.0);
    return s > 0.0 ? s : 120.0;
  }();
  return limit;
}

void spin_until(mcgpu_exchange* x, const std::atomic<long long>& counter, long long at_least, int peer, const char* what) {
  if (counter.load(std::memory_order_acquire) >= at_least) return;
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (counter.load(std::memory_order_acquire) < at_least) {
    if (x->boxes[peer].closing.load(std::memory_order_acquire) != 0)
      throw XError{-4, std::string("!!ERROR!! tally exchange: rank ") + std::to_string(peer) + " left while rank " + std::to_string(x->rank) + " waited for its " + what};
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 255) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_seconds())
      throw XError{-4, std::string("!!ERROR!! tally exchange: rank ") + std::to_string(x->rank) + " gave up waiting for the " + what + " of rank " + std::to_string(peer) +
                           " (MCGPU_EXCHANGE_TIMEOUT_S)"};
  }
  x->wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int fail(const XError& e) {
  mcgpu_set_last_error_(e.msg.c_str());
  return e.code ? e.code : -1;
}

}  // namespace

#define X_BEGIN try {
#define X_END                                   \
  }                                             \
  catch (const XError& e) { return fail(e); }   \
  catch (const std::exception& e) { return fail(XError{-2, e.what()}); }

extern "C" {

// + 64: the region is aligned up to a cache line here (the same offset in every process: mappings are page-aligned)
size_t mcgpu_exchange_shared_bytes(int world) { return (size_t)(world > 0 ? world : 1) * sizeof(Mailbox) + 64; }
size_t mcgpu_exchange_card_bytes(int world) { return (size_t)kHandle * (size_t)(1 + 2 + 2 * (world > 0 ? world : 1)); }

int mcgpu_exchange_create(int device_id, int rank, int world, size_t words, int policy, void* shared, mcgpu_exchange** out) {
  X_BEGIN
  X_REQUIRE(out && shared && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && words > 0 && policy >= 0 && policy <= 3,
            "!!ERROR!! mcgpu_exchange_create: bad argument (1 <= world <= 64, policy bits 0..3, shared region required)");
  mcgpu_exchange* x = new mcgpu_exchange;
  x->device = device_id; x->rank = rank; x->world = world; x->words = words; x->policy = policy & MCGPU_EXCHANGE_ROTATE;
  x->local = (policy & MCGPU_EXCHANGE_LOCAL) != 0;
  const unsigned int shared_event = hipEventDisableTiming | (x->local ? 0u : (unsigned int)hipEventInterprocess);
  x->boxes = reinterpret_cast<Mailbox*>(((uintptr_t)shared + 63) & ~(uintptr_t)63);
  try {
    X_HIP(hipSetDevice(device_id));
    X_HIP(hipStreamCreateWithFlags(&x->copy, hipStreamNonBlocking));
    X_HIP(hipStreamCreateWithFlags(&x->gather, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      X_HIP(hipEventCreateWithFlags(&x->landed[b], hipEventDisableTiming));
      X_HIP(hipEventCreateWithFlags(&x->added[b], hipEventDisableTiming));
      X_HIP(hipMalloc((void**)&x->tally[b], words * 8));
      X_HIP(hipMemset(x->tally[b], 0, words * 8));
      X_HIP(hipEventCreateWithFlags(&x->tracked[b], hipEventDisableTiming));
      X_HIP(hipEventCreateWithFlags(&x->pushed_local[b], hipEventDisableTiming));
      X_HIP(hipEventCreateWithFlags(&x->consumed[b], shared_event));
    }
    if (world > 1) {
      X_HIP(hipMalloc((void**)&x->landing, (size_t)world * 2 * words * 8));
      for (int q = 0; q < 2; ++q) {
        // the landing slots of parity q, peer after peer: what the fused add reads
        std::vector<const unsigned long long*> src;
        for (int r = 0; r < world; ++r)
          if (r != rank) src.push_back(x->slot(x->landing, r, q));
        X_HIP(hipMalloc((void**)&x->src_table[q], src.size() * sizeof(void*)));
        X_HIP(hipMemcpy((void*)x->src_table[q], src.data(), src.size() * sizeof(void*), hipMemcpyHostToDevice));
      }
      for (int o = 0; o < world; ++o)
        for (int q = 0; q < 2; ++q)
          if (o != rank) X_HIP(hipEventCreateWithFlags(&x->pushed[o][q], shared_event));
    }
    X_HIP(hipEventCreate(&x->t_push0)); X_HIP(hipEventCreate(&x->t_push1));
    X_HIP(hipEventCreate(&x->t_acc0)); X_HIP(hipEventCreate(&x->t_acc1));
    X_HIP(hipDeviceSynchronize());
  } catch (...) {
    mcgpu_exchange_destroy(x);
    throw;
  }
  x->connected[rank] = true;
  *out = x;
  return 0;
  X_END
}

// card = [landing memory handle][consumed q=0][consumed q=1][pushed -> owner 0: q=0, q=1][pushed -> owner 1: ...]...
int mcgpu_exchange_card(mcgpu_exchange* x, unsigned char* card, size_t card_bytes) {
  X_BEGIN
  X_REQUIRE(x && card && card_bytes >= mcgpu_exchange_card_bytes(x->world), "!!ERROR!! mcgpu_exchange_card: buffer too small");
  memset(card, 0, card_bytes);
  if (x->world == 1) return 0;
  X_REQUIRE(!x->local, "!!ERROR!! mcgpu_exchange_card: this exchange was created for ranks of one process (MCGPU_EXCHANGE_LOCAL)");
  X_HIP(hipSetDevice(x->device));
  hipIpcMemHandle_t mh;
  X_STAGE("IPC memory handle", x->rank, x->rank, hipIpcGetMemHandle(&mh, x->landing));
  memcpy(card, &mh, kHandle);
  for (int q = 0; q < 2; ++q) {
    hipIpcEventHandle_t eh;
    X_HIP(hipIpcGetEventHandle(&eh, x->consumed[q]));
    memcpy(card + (size_t)kHandle * (1 + q), &eh, kHandle);
  }
  for (int o = 0; o < x->world; ++o)
    for (int q = 0; q < 2; ++q) {
      if (o == x->rank) continue;
      hipIpcEventHandle_t eh;
      X_HIP(hipIpcGetEventHandle(&eh, x->pushed[o][q]));
      memcpy(card + (size_t)kHandle * (3 + 2 * o + q), &eh, kHandle);
    }
  return 0;
  X_END
}

int mcgpu_exchange_connect(mcgpu_exchange* x, int peer, const unsigned char* card, size_t card_bytes) {
  X_BEGIN
  X_REQUIRE(x && card && peer >= 0 && peer < x->world && peer != x->rank && card_bytes >= mcgpu_exchange_card_bytes(x->world) && !x->connected[peer],
            "!!ERROR!! mcgpu_exchange_connect: bad argument");
  X_REQUIRE(!x->local, "!!ERROR!! mcgpu_exchange_connect: this exchange was created for ranks of one process (MCGPU_EXCHANGE_LOCAL)");
  X_HIP(hipSetDevice(x->device));
  hipIpcMemHandle_t mh;
  memcpy(&mh, card, kHandle);
  void* p = nullptr;
  X_STAGE("IPC memory handle", x->rank, peer, hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess));
  x->peer_landing[peer] = (unsigned long long*)p;
  x->peer_mapped_ipc[peer] = true;
  for (int q = 0; q < 2; ++q) {
    hipIpcEventHandle_t eh;
    memcpy(&eh, card + (size_t)kHandle * (1 + q), kHandle);
    X_STAGE("IPC event handle", x->rank, peer, hipIpcOpenEventHandle(&x->peer_consumed[peer][q], eh));
    memcpy(&eh, card + (size_t)kHandle * (3 + 2 * x->rank + q), kHandle);
    X_STAGE("IPC event handle", x->rank, peer, hipIpcOpenEventHandle(&x->peer_pushed[peer][q], eh));
  }
  x->opened_events[peer] = true;
  x->connected[peer] = true;
  return 0;
  X_END
}

// ranks of one process (the contexts of mcgpu_run_scan_multi): events and landing memory are used directly
int mcgpu_exchange_connect_local(mcgpu_exchange* x, mcgpu_exchange* peer) {
  X_BEGIN
  X_REQUIRE(x && peer && x != peer && peer->world == x->world && peer->words == x->words && peer->rank != x->rank && !x->connected[peer->rank] &&
                peer->boxes == x->boxes && peer->policy == x->policy && x->local && peer->local,
            "!!ERROR!! mcgpu_exchange_connect_local: the two ends do not belong to one exchange");
  const int r = peer->rank;
  if (peer->device != x->device) {
    X_HIP(hipSetDevice(x->device));
    int can = 0;
    X_STAGE("peer access", x->rank, r, hipDeviceCanAccessPeer(&can, x->device, peer->device));
    X_REQUIRE(can, (std::string("!!ERROR!! tally exchange set-up, stage [peer access] rank ") + std::to_string(x->rank) + " (device " + std::to_string(x->device) +
                    ") -> peer " + std::to_string(r) + " (device " + std::to_string(peer->device) + "): hipDeviceCanAccessPeer says no").c_str());
    const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) X_STAGE("peer access", x->rank, r, e);
    (void)hipGetLastError();
  }
  x->peer_landing[r] = peer->landing;
  for (int q = 0; q < 2; ++q) {
    x->peer_consumed[r][q] = peer->consumed[q];
    x->peer_pushed[r][q] = peer->pushed[x->rank][q];
  }
  x->connected[r] = true;
  return 0;
  X_END
}

// After every peer is connected and BEFORE the first step: one small copy-engine transfer into this rank's slot of every
// peer's landing buffer, waited for -- the same call, copy kind and stream as the pushes of mcgpu_exchange_submit.  A platform on
// which that transfer cannot reach a peer (no copy-engine path between two devices, a mapping that only kernels may touch) fails
// here, where the caller can still agree with the other ranks on another route, instead of in the middle of a scan.  The bytes
// written are overwritten by the first real push into that slot.
int mcgpu_exchange_probe(mcgpu_exchange* x) {
  X_BEGIN
  X_REQUIRE(x, "!!ERROR!! mcgpu_exchange_probe: bad argument");
  X_REQUIRE(x->last_submitted < 0, "!!ERROR!! mcgpu_exchange_probe: call it before the first step");
  X_HIP(hipSetDevice(x->device));
  const size_t bytes = (size_t)std::min<unsigned long long>(x->words, 512ULL) * 8;
  for (int o = 0; o < x->world; ++o) {
    if (o == x->rank) continue;
    X_REQUIRE(x->connected[o], "!!ERROR!! mcgpu_exchange_probe: a peer is not connected");
    for (int q = 0; q < 2; ++q)
      X_STAGE("peer copy (copy engine)", x->rank, o, hipMemcpyAsync(x->slot(x->peer_landing[o], x->rank, q), x->tally[q], bytes, hipMemcpyDeviceToDeviceNoCU, x->copy));
    // waited for per peer: a transfer that fails asynchronously is reported with the peer it was going to
    X_STAGE("peer copy (copy engine)", x->rank, o, hipStreamSynchronize(x->copy));
  }
  return 0;
  X_END
}

int mcgpu_exchange_owner(const mcgpu_exchange* x, long long step) { return x ? x->owner(step) : -1; }

// The tally buffer of `step` (parity step & 1), zeroed on `stream`.  The buffer's previous user was step - 2: its push
// (non-owner) must have left it; its collect (owner) was issued on this stream by the caller before this call.
int mcgpu_exchange_begin(mcgpu_exchange* x, long long step, void* hip_stream, void** tally) {
  X_BEGIN
  X_REQUIRE(x && tally && step >= 0, "!!ERROR!! mcgpu_exchange_begin: bad argument");
  // the buffer is zeroed here: its previous user, step - 2, must be done with it on this stream -- submitted, and collected if
  // this rank owned it (a caller that lags its collects by more than one step would otherwise lose an unreduced tally silently)
  X_REQUIRE(step == x->last_submitted + 1, "!!ERROR!! mcgpu_exchange_begin: steps are consecutive (begin, launch, submit)");
  X_REQUIRE(step < 2 || x->owner(step - 2) != x->rank || x->last_collected >= step - 2,
            "!!ERROR!! mcgpu_exchange_begin: the tally of step - 2 has not been collected yet (its buffer would be zeroed)");
  X_HIP(hipSetDevice(x->device));
  const int b = (int)(step & 1);
  hipStream_t s = (hipStream_t)hip_stream;
  if (x->pushed_local_valid[b]) X_HIP(hipStreamWaitEvent(s, x->pushed_local[b], 0));
  X_HIP(hipMemsetAsync(x->tally[b], 0, x->words * 8, s));
  *tally = x->tally[b];
  return 0;
  X_END
}

// After the tracking launch of `step` has been enqueued on `stream`.  Non-owners: copy-engine push into the owner's landing
// slot (rank, parity of the owned index), behind the kernel and behind the owner's pass over that slot's previous content.
int mcgpu_exchange_submit(mcgpu_exchange* x, long long step, void* hip_stream) {
  X_BEGIN
  X_REQUIRE(x && step >= 0, "!!ERROR!! mcgpu_exchange_submit: bad argument");
  X_REQUIRE(step == x->last_submitted + 1, "!!ERROR!! mcgpu_exchange_submit: steps are consecutive, each submitted once (the counters of the protocol count them)");
  x->last_submitted = step;
  X_HIP(hipSetDevice(x->device));
  const int b = (int)(step & 1), o = x->owner(step);
  hipStream_t s = (hipStream_t)hip_stream;
  X_HIP(hipEventRecord(x->tracked[b], s));
  if (o == x->rank || x->world == 1) return 0;
  X_REQUIRE(x->connected[o], "!!ERROR!! mcgpu_exchange_submit: not connected to the owner of this step");
  const long long k = x->owned_index(step);
  const int q = (int)(k & 1);
  if (k >= 2) {  // the slot still holds index k - 2 until the owner's pass over it has been issued ...
    spin_until(x, x->boxes[o].collect_issued, k - 1, o, "collect");
    X_HIP(hipStreamWaitEvent(x->copy, x->peer_consumed[o][q], 0));  // ... and has run
  }
  X_HIP(hipStreamWaitEvent(x->copy, x->tracked[b], 0));
  X_HIP(hipEventRecord(x->t_push0, x->copy));
  X_HIP(hipMemcpyAsync(x->slot(x->peer_landing[o], x->rank, q), x->tally[b], x->words * 8, hipMemcpyDeviceToDeviceNoCU, x->copy));
  X_HIP(hipEventRecord(x->t_push1, x->copy));
  X_HIP(hipEventRecord(x->pushed[o][q], x->copy));
  X_HIP(hipEventRecord(x->pushed_local[b], x->copy));
  x->pushed_local_valid[b] = true;
  x->timed_push = true;
  ++x->pushes;
  x->boxes[x->rank].push_issued[o].fetch_add(1, std::memory_order_release);
  return 0;
  X_END
}

// Owner of `step`: on `stream` (the tracking stream, normally behind the launch of step + 1) wait for every peer's push
// and add the landed tallies to the own one in one pass; *reduced = the complete tally, valid until begin(step + 2).
// Other ranks: *reduced = NULL.
int mcgpu_exchange_collect(mcgpu_exchange* x, long long step, void* hip_stream, void** reduced) {
  X_BEGIN
  X_REQUIRE(x && step >= 0, "!!ERROR!! mcgpu_exchange_collect: bad argument");
  if (reduced) *reduced = nullptr;
  X_REQUIRE(step > x->last_collected && step <= x->last_submitted, "!!ERROR!! mcgpu_exchange_collect: a step is collected once, in order, after it was submitted");
  x->last_collected = step;
  if (x->owner(step) != x->rank) return 0;
  X_HIP(hipSetDevice(x->device));
  const int b = (int)(step & 1);
  hipStream_t s = (hipStream_t)hip_stream;
  if (x->world > 1) {
    const long long k = x->owned_index(step);
    const int q = (int)(k & 1);
    for (int r = 0; r < x->world; ++r) {
      if (r == x->rank) continue;
      X_REQUIRE(x->connected[r], "!!ERROR!! mcgpu_exchange_collect: a peer is not connected");
      spin_until(x, x->boxes[r].push_issued[x->rank], k + 1, r, "push");
      X_HIP(hipStreamWaitEvent(x->gather, x->peer_pushed[r][q], 0));
    }
    X_HIP(hipEventRecord(x->landed[q], x->gather));
    X_HIP(hipStreamWaitEvent(s, x->landed[q], 0));
    X_HIP(hipEventRecord(x->t_acc0, s));
    X_HIP(mcgpu::launch_accumulate_many(x->tally[b], x->src_table[q], x->world - 1, x->words, s));
    X_HIP(hipEventRecord(x->t_acc1, s));
    X_HIP(hipEventRecord(x->added[q], s));
    X_HIP(hipStreamWaitEvent(x->gather, x->added[q], 0));
    X_HIP(hipEventRecord(x->consumed[q], x->gather));
    x->timed_acc = true;
    x->boxes[x->rank].collect_issued.fetch_add(1, std::memory_order_release);
  }
  ++x->collects;
  if (reduced) *reduced = x->tally[b];
  return 0;
  X_END
}

// out[0] = duration of the last push [ms] (copy engine), out[1] = of the last fused add [ms], out[2] = pushes, out[3] =
// collects, out[4] = host seconds spent waiting for peers' counters, out[5] = payload bytes of one push.  Waits for both.
int mcgpu_exchange_stats(mcgpu_exchange* x, double out[6]) {
  X_BEGIN
  X_REQUIRE(x && out, "!!ERROR!! mcgpu_exchange_stats: null argument");
  X_HIP(hipSetDevice(x->device));
  float ms = 0.f;
  out[0] = out[1] = 0.0;
  if (x->timed_push) { X_HIP(hipEventSynchronize(x->t_push1)); X_HIP(hipEventElapsedTime(&ms, x->t_push0, x->t_push1)); out[0] = ms; }
  if (x->timed_acc) { X_HIP(hipEventSynchronize(x->t_acc1)); X_HIP(hipEventElapsedTime(&ms, x->t_acc0, x->t_acc1)); out[1] = ms; }
  out[2] = (double)x->pushes; out[3] = (double)x->collects; out[4] = x->wait_seconds; out[5] = (double)(x->words * 8);
  return 0;
  X_END
}

void mcgpu_exchange_destroy(mcgpu_exchange* x) {
  if (!x) return;
  if (x->boxes) x->boxes[x->rank].closing.store(1, std::memory_order_release);
  if (x->device >= 0) {
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r) {
      if (x->peer_mapped_ipc[r] && x->peer_landing[r]) (void)hipIpcCloseMemHandle(x->peer_landing[r]);
      for (int q = 0; q < 2; ++q) {
        if (x->opened_events[r] && x->peer_pushed[r][q]) (void)hipEventDestroy(x->peer_pushed[r][q]);
        if (x->opened_events[r] && x->peer_consumed[r][q]) (void)hipEventDestroy(x->peer_consumed[r][q]);
        if (x->pushed[r][q]) (void)hipEventDestroy(x->pushed[r][q]);
      }
    }
    for (int b = 0; b < 2; ++b) {
      if (x->tally[b]) (void)hipFree(x->tally[b]);
      if (x->tracked[b]) (void)hipEventDestroy(x->tracked[b]);
      if (x->pushed_local[b]) (void)hipEventDestroy(x->pushed_local[b]);
      if (x->consumed[b]) (void)hipEventDestroy(x->consumed[b]);
    }
    if (x->landing) (void)hipFree(x->landing);
    for (int q = 0; q < 2; ++q) {
      if (x->src_table[q]) (void)hipFree((void*)x->src_table[q]);
      if (x->landed[q]) (void)hipEventDestroy(x->landed[q]);
      if (x->added[q]) (void)hipEventDestroy(x->added[q]);
    }
    if (x->gather) (void)hipStreamDestroy(x->gather);
    for (hipEvent_t e : {x->t_push0, x->t_push1, x->t_acc0, x->t_acc1})
      if (e) (void)hipEventDestroy(e);
    if (x->copy) (void)hipStreamDestroy(x->copy);
  }
  delete x;
}

}  // extern "C"
