// How fast can ONE workgroup (8 waves) push 128 KB of 16-byte-per-lane global stores -- alone on the chip, and with every CU
// doing the same at once?  (The read-out of conv2x2_gdn512: 9.2 k cycles per 128 KB tile = 14 B/clk/CU.)
//      hipcc --offload-arch=gfx950 -O2 tools/micro/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void burst(uint4 *out, unsigned long long *clk, int tiles, int gap_iters) {
    const int tid = threadIdx.x;
    uint4 v = make_uint4(tid, blockIdx.x, 3u, 4u);
    float f = (float)tid;
    unsigned long long acc = 0;
    for (int t = 0; t < tiles; ++t) {
        uint4 *yo = out + ((long long)(t * gridDim.x + blockIdx.x)) * 8192 + tid;   // 128 KB per (tile, workgroup), contiguous
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < 16; ++r) yo[r * 512] = v;
        asm volatile("" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();   // all 16 stores ISSUED (not acknowledged)
        acc += t1 - t0;
        for (int g = 0; g < gap_iters; ++g) f = f * 1.0001f + 0.5f;   // "compute phase" between bursts
        v.x += (unsigned)f;
    }
    if ((tid & 63) == 0) clk[blockIdx.x * 8 + (tid >> 6)] = acc / tiles;
}

static void run(int grid, int tiles, int gap) {
    uint4 *d; unsigned long long *c;
    const size_t bytes = (size_t)grid * tiles * 131072;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&c, grid * 8 * sizeof(unsigned long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(burst, dim3(grid), dim3(512), 0, 0, d, c, tiles, gap);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(burst, dim3(grid), dim3(512), 0, 0, d, c, tiles, gap);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("grid %4d  tiles %3d  gap %6d: issue of 16 stores per wave: median %6llu  max %6llu memtime ticks (100 MHz: x ~22 for cycles); kernel %.3f ms = %.2f TB/s\n",
           grid, tiles, gap, h[h.size() / 2], h.back(), ms, bytes / ms / 1e9);
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    run(1, 32, 0);
    run(1, 32, 20000);
    run(8, 32, 20000);      // one workgroup per XCD
    run(32, 32, 20000);
    run(256, 32, 0);
    run(256, 32, 500);
    run(256, 32, 1000);
    run(256, 32, 2000);      // ~27 us per tile: the period of conv2x2_gdn512's tiles
    run(256, 32, 3000);
    run(256, 32, 5000);
    run(256, 32, 20000);
    return 0;
}
