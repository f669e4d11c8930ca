// Gradient-bucket all-reduce behind the C ABI: RCCL over xGMI, one communicator per process (= per GPU).
// Replaces what the reference gets from Lightning's DDP plugin (ex_audioset.py:488-489 -> torch DDP -> NCCL bucket
// all-reduce).  RCCL is resolved at RUN time (dlopen, preferring a copy the process has already loaded, e.g. torch's):
// libpasst_amd.so itself has no link-time dependency on it, and a single-GPU user never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
// RCCL is dlopen'ed at run time; its header is only needed for a handful of types.  A ROCm install without the RCCL
// development headers still builds the library: the declarations below are the stable NCCL 2 ABI.
#if __has_include(<rccl/rccl.h>) && !defined(PA_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7, ncclBfloat16 = 9 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGetVersion(int*);
ncclResult_t ncclCommCount(const ncclComm_t, int*);
ncclResult_t ncclCommUserRank(const ncclComm_t, int*);
ncclResult_t ncclCommCuDevice(const ncclComm_t, int*);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

#include <cstring>
#include <mutex>
#include <string>

#include "pa_common.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_once;
thread_local std::string t_comm_error;

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)             // a copy that is already mapped (torch ships one) wins: one RCCL per process
        if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
        if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) { g_rccl.error = std::string("cannot load librccl.so: ") + dlerror(); return; }
#define PA_SYM(f) g_rccl.f = (decltype(g_rccl.f))dlsym(g_rccl.handle, "nccl" #f); if (!g_rccl.f) g_rccl.error = "librccl.so lacks nccl" #f;
    PA_SYM(GetUniqueId) PA_SYM(CommInitRank) PA_SYM(AllReduce) PA_SYM(CommDestroy) PA_SYM(GetErrorString)
    PA_SYM(GetVersion) PA_SYM(CommCount) PA_SYM(CommUserRank) PA_SYM(CommCuDevice)
#undef PA_SYM
}

int comm_fail(const std::string& what) { t_comm_error = what; return PA_ECOMM; }
int check_nccl(ncclResult_t r, const char* call) {
    if (r == ncclSuccess) return PA_OK;
    return comm_fail(std::string(call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
}
int ready() {
    std::call_once(g_once, load_rccl);
    return g_rccl.error.empty() ? PA_OK : comm_fail(g_rccl.error);
}
}  // namespace

extern "C" const char* pa_comm_last_error(void) { return t_comm_error.c_str(); }

extern "C" int pa_comm_unique_id(void* id_out) {
    if (!id_out) return PA_EINVAL;
    if (int rc = ready()) return rc;
    static_assert(sizeof(ncclUniqueId) == PA_COMM_ID_BYTES, "PA_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    return check_nccl(g_rccl.GetUniqueId((ncclUniqueId*)id_out), "ncclGetUniqueId");
}

extern "C" int pa_comm_init(const void* id, int rank, int world, void** comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return PA_EINVAL;
    if (int rc = ready()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    if (int rc = check_nccl(g_rccl.CommInitRank(&c, world, uid, rank), "ncclCommInitRank")) return rc;
    *comm_out = (void*)c;
    return PA_OK;
}

extern "C" int pa_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, void* stream) {
    if (!comm || !buf || count <= 0) return PA_EINVAL;
    if (dtype != PA_F32 && dtype != PA_BF16) return PA_EINVAL;
    if (int rc = ready()) return rc;
    return check_nccl(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == PA_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, (ncclComm_t)comm,
                                      (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int pa_comm_info(void* comm, int* rccl_version, int* nranks, int* rank, int* device) {
    if (int rc = ready()) return rc;
    if (rccl_version) if (int rc = check_nccl(g_rccl.GetVersion(rccl_version), "ncclGetVersion")) return rc;
    if (!comm) return (nranks || rank || device) ? PA_EINVAL : PA_OK;      // version only: no communicator needed
    if (nranks) if (int rc = check_nccl(g_rccl.CommCount((ncclComm_t)comm, nranks), "ncclCommCount")) return rc;
    if (rank) if (int rc = check_nccl(g_rccl.CommUserRank((ncclComm_t)comm, rank), "ncclCommUserRank")) return rc;
    if (device) if (int rc = check_nccl(g_rccl.CommCuDevice((ncclComm_t)comm, device), "ncclCommCuDevice")) return rc;
    return PA_OK;
}

extern "C" int pa_comm_destroy(void* comm) {
    if (!comm) return PA_EINVAL;
    if (int rc = ready()) return rc;
    return check_nccl(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}
