// Diagnostics of libsc2amd.so (sc2_clock_probe: no product path calls it) and the kernel copy the host-coder batches of the stage
// pipeline cross PCIe with (sc2_copy_bytes).
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

// 16-byte copy between any two addresses the device can reach -- device memory or PINNED, device-mapped host memory (hipHostMalloc:
// what torch's pinned tensors are).  The stage pipeline's host-coder batches cross PCIe with it instead of hipMemcpyAsync:
// submitting a second copy-engine transfer while one is in flight blocked the CALLING thread for 5.7 ms on this runtime
// (profiles/r06k_copy_call_stall.txt), a kernel launch never waits for anything.  32 workgroups with four 16-byte transfers in flight
// per lane (512 KB on the link) fill PCIe and stay out of the way: with 512 workgroups resident on every CU for the ~1.4 ms a
// 74 MB batch takes to cross, the persistent encoder kernels -- which need a CU's whole register file -- could not be placed and the
// front stages beside the copy ran 3 - 6x slower (profiles/r06l_timeline.txt).
__global__ __launch_bounds__(256) void copy16_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int sc2_copy_bytes(void *dst, const void *src, long long n_bytes, void *stream) {
    SC2_REQUIRE(dst && src && n_bytes >= 0 && n_bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0, SC2_ERR_INVALID_ARG,
                "copy_bytes: 16-byte aligned pointers and a multiple of 16 bytes");
    if (n_bytes == 0) return SC2_OK;
    const long long n16 = n_bytes / 16;
    long long blocks = (n16 + 255) / 256;
    if (blocks > 32) blocks = 32;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<uint4 *>(dst),
                       static_cast<const uint4 *>(src), n16);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_clock_probe(unsigned long long *samples, int n_workgroups, int n_samples, unsigned period_ticks, void *stream) {
    SC2_REQUIRE(samples && n_workgroups > 0 && n_samples > 0 && period_ticks > 0, SC2_ERR_INVALID_ARG, "clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), samples, n_samples,
                       period_ticks);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
