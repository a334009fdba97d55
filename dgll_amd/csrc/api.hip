// api.hip -- library-level entry points and error plumbing of libdgll_hip.so.
#include <cstring>

#include "common.hpp"

namespace dgll {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();  // clear the sticky error so later calls can succeed
    return DGLL_ERR_HIP;
}

}  // namespace dgll

DGLL_API int dgll_hip_abi_version(void) { return DGLL_HIP_ABI_VERSION; }

DGLL_API const char* dgll_hip_last_error(void) { return dgll::g_last_error.c_str(); }

DGLL_API int dgll_hip_device_info(int device, char* name, int name_len, int* compute_units, int64_t* global_mem_bytes) {
    hipDeviceProp_t prop;
    DGLL_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        std::strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = '\0';
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = (int64_t)prop.totalGlobalMem;
    return DGLL_OK;
}
