// Root cause of the wait states behind `buf_store16` (conv2x2_win.hip, conv1x1_pair.hip): does a 16-byte buffer store still read
// its data registers after issue when its scalar offset is an SGPR?
//
// The documented hazard ("VMEM store of more than 64 bits followed by a VALU write of its data VGPRs") is handled by hipcc's
// hazard recogniser -- but for MUBUF stores only when the soffset field is NOT a register (GCNHazardRecognizer exempts stores
// with an SGPR soffset).  The kernels' stores all carry an SGPR soffset.  This program issues, in ONE asm statement so that the
// compiler cannot interfere,
//      buffer_store_dwordx4 v[10:13], voff, rsrc, <soffset>        (data = A)
//      [s_nop NOPS]
//      v_mov_b32 v10..v13, B
// on 8 waves per CU (two per SIMD) and counts stored elements that are not A.
//
//      hipcc --offload-arch=gfx950 -O2 tools/micro/store_hazard.hip -o /tmp/store_hazard && /tmp/store_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;

template <int NOPS, bool SGPR_SOFF>
__global__ __launch_bounds__(512) void store_then_write(uint32_t *out, int iters, uint32_t bytes) {
    const buf_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)bytes, 0x00020000);
    const uint32_t lane_slot = (blockIdx.x * 512u + threadIdx.x) * 16u;   // one 16-byte slot per thread and iteration row
    const uint32_t row_bytes = gridDim.x * 512u * 16u;
    const uint32_t A = 0xAAAA0000u | threadIdx.x, B = 0xBBBBBBBBu;
    for (int i = 0; i < iters; ++i) {
        const uint32_t voff = lane_slot;
        const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)i * row_bytes);
#define SC2_BODY(SOFF_OPERAND, NOP_TEXT)                                                                                           \
    asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %2\n\ts_nop 4\n\t"               \
                 "buffer_store_dwordx4 v[10:13], %0, %1, " SOFF_OPERAND " offen\n\t" NOP_TEXT                                      \
                 "v_mov_b32 v10, %4\n\tv_mov_b32 v11, %4\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %4\n\t"                           \
                 :: "v"(voff), "s"(r), "v"(A), "s"(soff), "v"(B) : "v10", "v11", "v12", "v13", "memory")
        if (SGPR_SOFF) {
            if (NOPS == 0) SC2_BODY("%3", "");
            else if (NOPS == 1) SC2_BODY("%3", "s_nop 0\n\t");
            else if (NOPS == 2) SC2_BODY("%3", "s_nop 1\n\t");
            else SC2_BODY("%3", "s_nop 7\n\t");
        } else {
            const uint32_t voff2 = voff + soff;
            asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %2\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[10:13], %0, %1, 0 offen\n\t"
                         "v_mov_b32 v10, %3\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %3\n\t"
                         :: "v"(voff2), "s"(r), "v"(A), "v"(B) : "v10", "v11", "v12", "v13", "memory");
        }
    }
}

template <int NOPS, bool SGPR_SOFF>
void run(const char *name) {
    const int blocks = 256, iters = 64;
    const size_t n = (size_t)blocks * 512 * 4 * iters;
    uint32_t *d;
    (void)hipMalloc(&d, n * 4);
    long long bad = 0, bad_lanes[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL((store_then_write<NOPS, SGPR_SOFF>), dim3(blocks), dim3(512), 0, 0, d, iters, (uint32_t)(n * 4));
        (void)hipDeviceSynchronize();
        std::vector<uint32_t> h(n);
        (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t tid = (uint32_t)((i / 4) % 512);
            if (h[i] != (0xAAAA0000u | tid)) {
                ++bad;
                ++bad_lanes[(tid & 15) >> 2];
            }
        }
    }
    printf("%-52s wrong elements %lld of %zu  (by lane quarter of 16: %lld %lld %lld %lld)\n", name, bad, n * 8, bad_lanes[0],
           bad_lanes[1], bad_lanes[2], bad_lanes[3]);
    (void)hipFree(d);
}

int main() {
    run<0, true>("SGPR soffset, VALU write directly behind the store");
    run<1, true>("SGPR soffset, s_nop 0");
    run<2, true>("SGPR soffset, s_nop 1");
    run<3, true>("SGPR soffset, s_nop 7");
    run<0, false>("soffset 0 (literal), VALU write directly behind");
    return 0;
}
