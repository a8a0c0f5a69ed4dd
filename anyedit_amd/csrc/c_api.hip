// libanyedit_hip.so — C-ABI plumbing: version, error string, device query.
// Conventions (include/anyedit_hip.h): every entry point returns int (0 = ok, <0 = error), never throws,
// never allocates caller-visible memory, never synchronises; pointers are device pointers borrowed for the call.
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void ae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ae_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ae_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return AE_ERR_LAUNCH;
    }
    return AE_OK;
}

extern "C" int ae_version(void) { return 100; }  // 0.1.0

extern "C" const char* ae_last_error(void) { return g_err; }

extern "C" int ae_device_arch(char* buf, int n) {
    if (!buf || n <= 0) return AE_ERR_ARG;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        ae_set_error("ae_device_arch: no HIP device");
        buf[0] = 0;
        return AE_ERR_LAUNCH;
    }
    strncpy(buf, prop.gcnArchName, (size_t)n - 1);
    buf[n - 1] = 0;
    return AE_OK;
}

extern "C" int ae_device_info(int* cus, long* hbm_bytes, int* clock_khz) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        ae_set_error("ae_device_info: no HIP device");
        return AE_ERR_LAUNCH;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (long)prop.totalGlobalMem;
    if (clock_khz) *clock_khz = prop.clockRate;
    return AE_OK;
}
