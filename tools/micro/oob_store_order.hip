// Does a vector-memory STORE that the buffer descriptor's range check drops keep its place in the order in which vmcnt retires?
//
// The kernels with hand-counted `s_waitcnt vmcnt(N)` rely on in-order retirement: "at most N operations outstanding" means every
// operation older than the N youngest has completed.  Several of them mask lanes (or whole instructions: the rows past the last
// tile) by sending the store's offset out of the descriptor's range.  If such an instruction were acknowledged early -- it moves no
// data -- a wait that counts it among the younger operations would end while an OLDER load is still in flight.
//
// Test, in one asm statement per iteration so that the compiler cannot interfere:
//      v_mov   vD, SENTINEL
//      buffer_load_dword  vD, voff_far, rsrc_in          (a cold line far away: long latency)
//      N x buffer_store_dword  vS, voff_store, rsrc_out  (voff_store out of range, or in range: the control)
//      s_waitcnt vmcnt(N)                                (by the in-order model: the load has landed)
//      v_mov   vR, vD
// and vR is compared with what the load must return.  A SENTINEL in vR = the wait ended before the load landed.  Control: the same
// with vmcnt(9), which by construction does not wait for the load -- the probe must report sentinels there.
//
//      hipcc --offload-arch=gfx950 -O2 tools/micro/oob_store_order.hip -o /tmp/oob_store_order && /tmp/oob_store_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;

constexpr uint32_t SENTINEL = 0xDEADBEEFu;

template <int N, bool OOB, int WAIT>
__global__ __launch_bounds__(256) void probe(const uint32_t *in, uint32_t in_bytes, uint32_t *sink, uint32_t sink_bytes, uint32_t *got,
                                             int iters, uint32_t stride_words) {
    const buf_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(in), 0, (int)in_bytes, 0x00020000);
    const buf_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(sink, 0, (int)sink_bytes, 0x00020000);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t bad = 0;
    for (int i = 0; i < iters; ++i) {
        // a different cold 128-byte line per thread and iteration
        const uint32_t w = (t * 32u + (uint32_t)i * stride_words) % (in_bytes / 4u);
        const uint32_t voff = w * 4u;
        const uint32_t soff = OOB ? 0x80000000u : (t * 4u) % sink_bytes;
        uint32_t dst, res;   // dst: the load's destination; res: its value copied right behind the wait (a later arrival cannot change it)
#define ST "buffer_store_dword %4, %5, %6, 0 offen\n\t"
#define ST4 ST ST ST ST
        asm volatile("v_mov_b32 %0, %7\n\t"
                     "s_nop 4\n\t"
                     "buffer_load_dword %0, %2, %3, 0 offen\n\t"
                     ST4 ST4
                     "s_waitcnt vmcnt(%8)\n\t"
                     "v_mov_b32 %1, %0\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     : "=&v"(dst), "=&v"(res)
                     : "v"(voff), "s"(r_in), "v"(t), "v"(soff), "s"(r_out), "v"(SENTINEL), "n"(WAIT)
                     : "memory");
        (void)dst;
        static_assert(N == 8, "eight stores behind the load");
        bad += res != in[w] ? 1u : 0u;
        bad += res == SENTINEL ? 0x10000u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    got[t] = bad;
}

template <bool OOB, int WAIT>
void run(const char *name, const uint32_t *d_in, uint32_t in_bytes, uint32_t *d_sink, uint32_t sink_bytes) {
    const int blocks = 2048, iters = 256;
    uint32_t *d_got;
    (void)hipMalloc(&d_got, blocks * 256 * 4);
    long long wrong = 0, sentinels = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(d_got, 0, blocks * 256 * 4);
        hipLaunchKernelGGL((probe<8, OOB, WAIT>), dim3(blocks), dim3(256), 0, 0, d_in, in_bytes, d_sink, sink_bytes, d_got, iters,
                           (uint32_t)(7919u * 32u * (rep + 1)));
        (void)hipDeviceSynchronize();
        std::vector<uint32_t> h(blocks * 256);
        (void)hipMemcpy(h.data(), d_got, h.size() * 4, hipMemcpyDeviceToHost);
        for (uint32_t v : h) {
            wrong += v & 0xFFFFu;
            sentinels += v >> 16;
        }
    }
    printf("%-44s probes %lld   wrong values %lld   of them the sentinel (wait ended before the load landed) %lld\n", name,
           4LL * blocks * 256 * iters, wrong, sentinels);
    (void)hipFree(d_got);
}

int main() {
    const uint32_t in_bytes = 1u << 30;   // 1 GB: every probed line is cold
    const uint32_t sink_bytes = 1u << 20;
    uint32_t *d_in, *d_sink;
    if (hipMalloc(&d_in, in_bytes) != hipSuccess || hipMalloc(&d_sink, sink_bytes) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    std::vector<uint32_t> h(in_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;   // never the sentinel's pattern by construction below
    for (size_t i = 0; i < h.size(); ++i)
        if (h[i] == SENTINEL) h[i] ^= 2u;
    (void)hipMemcpy(d_in, h.data(), in_bytes, hipMemcpyHostToDevice);
    (void)hipMemset(d_sink, 0, sink_bytes);
    run<false, 9>("control: in-range stores, vmcnt(9) (too weak)", d_in, in_bytes, d_sink, sink_bytes);   // the probe must SEE an early wait
    run<false, 8>("8 in-range stores behind the load, vmcnt(8)", d_in, in_bytes, d_sink, sink_bytes);
    run<true, 8>("8 OUT-OF-RANGE stores behind the load, vmcnt(8)", d_in, in_bytes, d_sink, sink_bytes);
    return 0;
}
