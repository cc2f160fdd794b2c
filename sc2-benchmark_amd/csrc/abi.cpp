// Library-level entry points of libsc2amd.so: ABI version, error string, device probe.
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/sc2_bottleneck.h"

namespace {
thread_local char g_err[512] = "";
}

void sc2_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sc2_abi_version(void) { return SC2_ABI_VERSION; }
extern "C" const char *sc2_last_error(void) { return g_err; }
extern "C" int sc2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
