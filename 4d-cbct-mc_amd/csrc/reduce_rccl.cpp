// reduce_rccl.cpp -- the vendor-collective route of the multi-device scan: north_star's "single RCCL reduce over xGMI on the
// per-projection detector image", i.e. the reference's MPI_Reduce (docker/mcgpu/MC-GPU_v1.3.cu:1006-1024: every rank's 45 MB
// tally to the host, one blocking MPI_Reduce(MPI_UNSIGNED_LONG_LONG, MPI_SUM, root 0) per projection) as one
// ncclReduce(uint64, sum, root) per projection between the devices of THIS process (the drop-in executable drives all devices of
// a node from one process: ncclCommInitAll).
//
// RCCL is NOT linked: the library is opened on first use (dlopen "librccl.so.1"), so the engine loads and runs on a machine
// without it and a one-device run never pays for it.  scan.cpp takes this route when asked to (MCGPU_REDUCE=rccl, `--reduce rccl`)
// or when the tally exchange is unavailable between the devices, before it falls back to projection sharding.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mcgpu_amd.h"
#include "knobs.hpp"

extern "C" void mcgpu_set_last_error_(const char* message);  // engine.cpp

namespace {
// the few declarations of <rccl/rccl.h> this file needs (the header itself may be absent where the library is)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclSuccess = 0, kNcclUint64 = 5, kNcclSum = 0;  // rccl.h: ncclSuccess, ncclUint64 (ncclDataType_t), ncclSum (ncclRedOp_t)
typedef ncclResult_t (*CommInitAllFn)(ncclComm_t*, int, const int*);
typedef ncclResult_t (*CommDestroyFn)(ncclComm_t);
typedef ncclResult_t (*ReduceFn)(const void*, void*, size_t, int, int, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*GroupFn)();
typedef const char* (*ErrorStringFn)(ncclResult_t);
}  // namespace

struct mcgpu_rccl {
  void* lib = nullptr;
  int n = 0;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  CommDestroyFn comm_destroy = nullptr;
  ReduceFn reduce = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  ErrorStringFn error_string = nullptr;
};

namespace {
int fail(const std::string& msg) {
  mcgpu_set_last_error_(msg.c_str());
  return -1;
}
}  // namespace

extern "C" {

// One communicator per device of `devices` (ncclCommInitAll).  Returns 0, or -1 with mcgpu_last_error() = why this route cannot be
// taken here (no library, a device listed twice, no path between the devices ...): the caller falls back.
int mcgpu_rccl_create(const int* devices, int n, mcgpu_rccl** out) {
  if (!devices || n < 1 || !out) return fail("!!ERROR!! mcgpu_rccl_create: bad argument");
  *out = nullptr;
  if (mcgpu::knob_set("MCGPU_RCCL_FAIL")) return fail("!!ERROR!! RCCL reduction: failure requested (MCGPU_RCCL_FAIL)");  // test hook of the fallback chain
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (devices[a] == devices[b]) return fail("!!ERROR!! RCCL reduction: a device is listed twice (RCCL wants one rank per GPU)");
  const char* name = mcgpu::knob_str("MCGPU_RCCL_LIBRARY");
  void* lib = dlopen(name ? name : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib && !name) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    const char* why = dlerror();  // may be NULL (no pending message): never hand that to std::string
    return fail(std::string("!!ERROR!! RCCL reduction: cannot open the RCCL library (") + (why ? why : "no reason given by dlopen") + ")");
  }
  mcgpu_rccl* r = new mcgpu_rccl;
  r->lib = lib;
  r->n = n;
  r->devices.assign(devices, devices + n);
  CommInitAllFn init_all = (CommInitAllFn)dlsym(lib, "ncclCommInitAll");
  r->comm_destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
  r->reduce = (ReduceFn)dlsym(lib, "ncclReduce");
  r->group_start = (GroupFn)dlsym(lib, "ncclGroupStart");
  r->group_end = (GroupFn)dlsym(lib, "ncclGroupEnd");
  r->error_string = (ErrorStringFn)dlsym(lib, "ncclGetErrorString");
  if (!init_all || !r->comm_destroy || !r->reduce || !r->group_start || !r->group_end) {
    mcgpu_rccl_destroy(r);
    return fail("!!ERROR!! RCCL reduction: the library lacks ncclCommInitAll / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclCommDestroy");
  }
  r->comms.assign((size_t)n, nullptr);
  const ncclResult_t rc = init_all(r->comms.data(), n, devices);
  if (rc != kNcclSuccess) {
    const std::string why = r->error_string ? r->error_string(rc) : "ncclCommInitAll failed";
    r->comms.clear();
    mcgpu_rccl_destroy(r);
    return fail("!!ERROR!! RCCL reduction: ncclCommInitAll: " + why);
  }
  *out = r;
  return 0;
}

// tallies[g] (uint64[words] on device g) summed into tallies[root], in place on the root, on streams[g]: one ncclReduce per device
// inside one group (one thread drives every device of the node).
int mcgpu_rccl_reduce_u64(mcgpu_rccl* r, void* const* tallies, size_t words, int root, void* const* hip_streams) {
  if (!r || !tallies || !hip_streams || root < 0 || root >= r->n) return fail("!!ERROR!! mcgpu_rccl_reduce_u64: bad argument");
  auto why = [&](ncclResult_t rc) { return std::string((r->error_string && rc > 0) ? r->error_string(rc) : "failed"); };
  ncclResult_t rc = r->group_start();
  if (rc != kNcclSuccess) return fail("!!ERROR!! RCCL reduction: ncclGroupStart: " + why(rc));  // no group is open: nothing to end
  const char* step = "ncclReduce";
  for (int g = 0; g < r->n && rc == kNcclSuccess; ++g) {
    if (hipSetDevice(r->devices[(size_t)g]) != hipSuccess) { rc = -1; step = "hipSetDevice"; break; }
    rc = r->reduce(tallies[g], tallies[g], words, kNcclUint64, kNcclSum, root, r->comms[(size_t)g], (hipStream_t)hip_streams[g]);
  }
  // the group was opened: it is closed whatever happened inside it (RCCL keeps per-thread group state), and the FIRST failure is reported
  const ncclResult_t rc_end = r->group_end();
  if (rc != kNcclSuccess) return fail(std::string("!!ERROR!! RCCL reduction: ") + step + ": " + why(rc));
  if (rc_end != kNcclSuccess) return fail("!!ERROR!! RCCL reduction: ncclGroupEnd: " + why(rc_end));
  return 0;
}

void mcgpu_rccl_destroy(mcgpu_rccl* r) {
  if (!r) return;
  for (size_t g = 0; g < r->comms.size(); ++g)
    if (r->comms[g] && r->comm_destroy) {
      (void)hipSetDevice(r->devices[g]);
      (void)r->comm_destroy(r->comms[g]);
    }
  if (r->lib) dlclose(r->lib);
  delete r;
}

}  // extern "C"
