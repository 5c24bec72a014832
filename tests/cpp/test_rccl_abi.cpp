// Compile-time check: cddp-cpp_amd/csrc/comm.hip restates the few RCCL declarations it needs so that the library builds
// without the RCCL headers and resolves librccl with dlopen.  Where the headers ARE installed this translation unit holds those restatements to
// them: the opaque id size, the two enum values, and the signatures of the seven entry points comm.hip binds by name.  Compile only (-fsyntax-only).
#include <rccl/rccl.h>
#include <type_traits>

static_assert(sizeof(ncclUniqueId) == 128, "comm.hip: ncclUniqueId is restated as 128 opaque bytes");
#include "../../include/cddp_hip.h"
static_assert(NCCL_UNIQUE_ID_BYTES == CDDP_HIP_COMM_ID_BYTES, "cddp_hip.h: CDDP_HIP_COMM_ID_BYTES is the size of the id cddp_hip_comm_unique_id hands out");
static_assert((int)ncclSuccess == 0, "comm.hip: ncclSuccess == 0");
static_assert((int)ncclUint8 == 1, "comm.hip: ncclUint8 == 1");
static_assert(std::is_pointer<ncclComm_t>::value, "comm.hip: ncclComm_t is an opaque pointer");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int), "comm.hip declares both enums as int");

static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclCommCount), ncclResult_t (*)(const ncclComm_t, int *)>::value, "ncclCommCount");
static_assert(std::is_same<decltype(&ncclCommUserRank), ncclResult_t (*)(const ncclComm_t, int *)>::value, "ncclCommUserRank");
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "ncclGetErrorString");
int main() { return 0; }
