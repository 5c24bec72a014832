// The single collective of the path: one RCCL all-gather of the 16-byte {final_cost, iterations,
// status} records per batch solve, at the C-ABI so that a C++ host has the multi-GPU path without
// Python.  RCCL is resolved lazily (dlopen) -- the solver core itself has no link-time dependency on it, and a process
// that already loaded an RCCL (PyTorch bundles one) keeps using that copy.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include <hip/hip_runtime.h>

// The few RCCL declarations this file needs, restated so that the solver core builds on a box without the RCCL headers
// (the library is resolved with dlopen at first use; cddp_hip_comm_* return -30 when it is absent).  They follow the
// stable NCCL 2.x C interface: ncclUniqueId is 128 opaque bytes, ncclComm_t an opaque pointer, ncclSuccess == 0,
// ncclUint8 == 1 in ncclDataType_t.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1;

#include "../../include/cddp_hip.h"

extern "C" int cddp_hip_internal_set_error(int code, const char *msg);   // capi.hip (thread-local last-error string)

namespace {

int cfail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  return cddp_hip_internal_set_error(code, buf);
}

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  // First the librccl that sits next to the HIP runtime THIS library is bound to: RCCL opens the HSA runtime of its own
  // directory, and a process that holds two ROCm installations (PyTorch bundles one) must not mix them -- a communicator on
  // the other copy fails with "no ROCm-capable device" (seen when this library initialised /opt/rocm's runtime before torch
  // was imported).  With torch imported first both resolve to torch's copy; see INTEGRATION.md section 4.
  {
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char *n : {"librccl.so.1", "librccl.so"}) { r.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
      }
    }
  }
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  if (!r.lib) for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
  if (!r.lib) { r.err = std::string("cannot load librccl: ") + dlerror(); return r; }
  auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.err = std::string("librccl lacks ") + n; return p; };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  return r;
}

}  // namespace

extern "C" {

int cddp_hip_comm_unique_id(char *id_out) {
  if (!id_out) return cfail(-1, "null argument");
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(-30, "%s", r.err.c_str());
  ncclUniqueId id;
  ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) return cfail(-31, "ncclGetUniqueId: %s", r.GetErrorString(rc));
  static_assert(sizeof(id) == CDDP_HIP_COMM_ID_BYTES, "unique id size");
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int cddp_hip_comm_init(const char *id_in, int world, int rank, int device, void **comm_out) {
  if (!id_in || !comm_out) return cfail(-1, "null argument");
  if (world <= 0 || rank < 0 || rank >= world) return cfail(-1, "bad rank %d of world %d", rank, world);
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(-30, "%s", r.err.c_str());
  if (hipSetDevice(device) != hipSuccess) return cfail(-10, "hipSetDevice(%d) failed", device);
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t rc = r.CommInitRank(&c, world, id, rank);
  if (rc != ncclSuccess) return cfail(-31, "ncclCommInitRank: %s", r.GetErrorString(rc));
  *comm_out = (void *)c;
  return 0;
}

int cddp_hip_comm_destroy(void *comm) {
  if (!comm) return 0;
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(-30, "%s", r.err.c_str());
  ncclResult_t rc = r.CommDestroy((ncclComm_t)comm);
  if (rc != ncclSuccess) return cfail(-31, "ncclCommDestroy: %s", r.GetErrorString(rc));
  return 0;
}

// How many ranks RCCL itself sees behind a communicator, and which one this is (ncclCommCount / ncclCommUserRank): lets a launcher
// assert that the exchange really spans `world` processes instead of trusting its own bookkeeping.
int cddp_hip_comm_info(void *comm, int *count_out, int *rank_out) {
  if (!comm || !count_out || !rank_out) return cfail(-1, "null argument");
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(-30, "%s", r.err.c_str());
  ncclResult_t rc = r.CommCount((ncclComm_t)comm, count_out);
  if (rc != ncclSuccess) return cfail(-31, "ncclCommCount: %s", r.GetErrorString(rc));
  rc = r.CommUserRank((ncclComm_t)comm, rank_out);
  if (rc != ncclSuccess) return cfail(-31, "ncclCommUserRank: %s", r.GetErrorString(rc));
  return 0;
}

// used by capi.hip::cddp_hip_allgather_results
int cddp_hip_internal_allgather(const void *send, void *recv, size_t bytes_per_rank, void *comm, void *stream) {
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(-30, "%s", r.err.c_str());
  ncclResult_t rc = r.AllGather(send, recv, bytes_per_rank, ncclUint8, (ncclComm_t)comm, (hipStream_t)stream);
  if (rc != ncclSuccess) return cfail(-31, "ncclAllGather: %s", r.GetErrorString(rc));
  return 0;
}

}  // extern "C"
