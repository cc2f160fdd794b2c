// Diagnostics of libsc2amd.so (no product path calls these).
//
// sc2_clock_probe: the shader clock the chip HOLDS while other kernels run (MI355X_MICROARCH.md, "DVFS give-back" item 6: the
// in-kernel clock is delta s_memtime / delta s_memrealtime x 100 MHz).  Instead of stamping every kernel whose clock is wanted in a
// build of its own, ONE probe wave per workgroup spins on the constant 100 MHz counter and writes (s_memtime, s_memrealtime, XCC id)
// every `period` ticks while the kernel under study runs beside it on another stream: s_memtime counts the cycles of the XCD's
// shader clock, whoever keeps the XCD busy.  A probe wave is a scalar loop -- one instruction stream on one SIMD, no LDS, 8
// registers -- so it fits beside anything but a kernel that owns every register of every SIMD of its CU, and then it takes that
// CU's place in the dispatch order (launch it first).  tools/clock_probe.py turns the samples into per-XCD clocks and, with the
// kernel's own duration and FLOP count, into the fraction of matrix-pipe cycles it uses at THAT clock.
#include "sc2_common.h"

namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, int n_samples, unsigned period) {
    if (threadIdx.x != 0) return;
    unsigned long long *row = out + (size_t)blockIdx.x * (size_t)n_samples * 3u;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu;   // HW_REG_XCC_ID, bits 3:0
    unsigned long long r = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_samples; ++i) {
        const unsigned long long next = r + period;
        do {
            __builtin_amdgcn_s_sleep(8);
            r = __builtin_amdgcn_s_memrealtime();
        } while (r < next);
        row[3 * i + 0] = __builtin_amdgcn_s_memtime();
        row[3 * i + 1] = r;
        row[3 * i + 2] = xcc;
    }
}

}  // namespace

extern "C" int sc2_clock_probe(unsigned long long *samples, int n_workgroups, int n_samples, unsigned period_ticks, void *stream) {
    SC2_REQUIRE(samples && n_workgroups > 0 && n_samples > 0 && period_ticks > 0, SC2_ERR_INVALID_ARG, "clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), samples, n_samples,
                       period_ticks);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
