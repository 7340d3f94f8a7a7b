// Error reporting, version and device queries of libsnuffy_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace snf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SNF_ELAUNCH;
    }
    return SNF_OK;
}

int cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local int cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;  // MI355X default; only reached without a device
    if (dev != cached_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached = v;
        cached_dev = dev;
    }
    return cached;
}

// one bit per device id (ids >= 63 get no bit: their opt-ins are simply repeated on every launch): the key of the per-thread "this kernel has
// its LDS opt-in on this device" masks -- hipFuncSetAttribute is per device, a thread that moves to another GPU has to repeat it
unsigned long long device_bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev >= 63 ? 0ull : 1ull << dev;     // 0 = never remembered: always set
}

}  // namespace snf

extern "C" {

const char* snf_version(void) { return "snuffy_hip 0.3.0 (gfx950)"; }
const char* snf_last_error(void) { return snf::g_err; }
int snf_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        snf::set_error("snf_device_cu_count: no HIP device");
        (void)hipGetLastError();
        return SNF_ELAUNCH;
    }
    return snf::cu_count();
}

}  // extern "C"
