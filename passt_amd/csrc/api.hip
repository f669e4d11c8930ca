// Error reporting + ABI version of libpasst_amd.so.
#include <string>

#include "pa_common.h"

namespace pa {
static thread_local std::string g_last_hip_error;
int set_hip_error(hipError_t e) {
    g_last_hip_error = hipGetErrorString(e);
    return PA_ELAUNCH;
}
}  // namespace pa

extern "C" int pa_abi_version(void) { return PA_ABI_VERSION; }

extern "C" const char* pa_error_string(int code) {
    switch (code) {
        case PA_OK: return "ok";
        case PA_EINVAL: return "invalid argument";
        case PA_EUNSUPPORTED: return "unsupported shape or dtype";
        case PA_ELAUNCH: return "kernel launch failed (see pa_last_hip_error)";
        case PA_ECOMM: return "RCCL unavailable or collective failed (see pa_comm_last_error)";
    }
    return "unknown error code";
}

extern "C" const char* pa_last_hip_error(void) { return pa::g_last_hip_error.c_str(); }
