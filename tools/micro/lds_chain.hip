// Latency of DEPENDENT LDS reads on a lone wave (the range coder's per-symbol chain holds two): ds_read_u8 -> address -> ds_read_b32.
//      hipcc --offload-arch=gfx950 -O2 tools/micro/lds_chain.hip -o /tmp/lds_chain && /tmp/lds_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(64) void chain(unsigned *out, unsigned long long *clk, int n) {
    __shared__ unsigned tab[16384];
    __shared__ unsigned char tab8[65536];
    for (int i = threadIdx.x; i < 16384; i += 64) tab[i] = (i * 2654435761u) & 16383u;
    for (int i = threadIdx.x; i < 65536; i += 64) tab8[i] = (unsigned char)(i * 40503u >> 3);
    __syncthreads();
    unsigned a = threadIdx.x * 37u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) a = tab[a & 16383u];                                   // one dword trip
            else if (KIND == 1) a = tab[tab8[a & 65535u] * 4u + (a & 3u)];         // byte trip -> dword trip (the LUT decoder's pair)
            else a = tab8[a & 65535u] * 257u + (a >> 3);                          // one byte trip
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = a;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char *name) {
    unsigned *out; unsigned long long *clk;
    (void)hipMalloc(&out, 64 * 4); (void)hipMalloc(&clk, 8);
    hipLaunchKernelGGL(chain<KIND>, dim3(1), dim3(64), 0, 0, out, clk, 2000);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(chain<KIND>, dim3(1), dim3(64), 0, 0, out, clk, 20000);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f s_memtime ticks per step\n", name, (double)h / (8.0 * 20000));
}
int main() {
    run<0>("ds_read_b32 -> and -> ds_read_b32 ...");
    run<2>("ds_read_u8 -> mul/add/and -> ds_read_u8 ...");
    run<1>("ds_read_u8 -> ds_read_b32 (two trips per step)");
    return 0;
}
