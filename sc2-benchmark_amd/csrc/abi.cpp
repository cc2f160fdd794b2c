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

// ---- dispatch policy (include/sc2_bottleneck.h): one plain struct, no environment variable anywhere in the library
namespace {
sc2_policy make_default_policy() {
    sc2_policy p = {};
    p.struct_bytes = (int32_t)sizeof(sc2_policy);
    p.conv_patch3 = 1;
    p.conv_s2 = 1;
    p.conv_persist = 3;
    p.win_half = 1;
    p.pair_alt = 1;
    p.f32_persist0 = 1;
    p.rans_lds_pad_kb = 159;
    p.rans_pad_waves = 16;
    p.rans_ragged2 = 1;
    p.rans_ragged2_waves = 0;   // by stream count (rans.hip)
    p.rans_lut8 = 0;   // measured (round 5, K = 20 / 100 on one box): decode 12.1 -> 12.9 ms / 12.6 -> 13.4 ms with the one-lookup table
    p.rans_dq_lds = 0;
    p.wgrad_ct = 0;
    return p;
}
sc2_policy g_policy = make_default_policy();
}  // namespace

const sc2_policy &sc2_pol() { return g_policy; }

extern "C" void sc2_policy_default(sc2_policy *p) {
    if (p) *p = make_default_policy();
}
extern "C" int sc2_policy_set(const sc2_policy *p) {
    if (!p || p->struct_bytes != (int32_t)sizeof(sc2_policy)) {
        sc2_set_error("sc2_policy_set: null policy or struct_bytes %d != %d (header / library mismatch)", p ? p->struct_bytes : -1,
                      (int)sizeof(sc2_policy));
        return SC2_ERR_INVALID_ARG;
    }
    g_policy = *p;
    return SC2_OK;
}
extern "C" void sc2_policy_get(sc2_policy *p) {
    if (p) *p = g_policy;
}
