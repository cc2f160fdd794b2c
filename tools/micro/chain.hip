// micro-benchmark: latency of dependent instruction chains on ONE wave (what a serial coder sees), and the
// shader clock the chip holds while only a few waves are busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void chain(double *out, unsigned long long *clk, int n, double seed, unsigned useed) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    double a = seed + threadIdx.x, b = 1.0000001;
    unsigned u = useed + threadIdx.x;
    float f = (float)seed;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { a = fma(a, b, 1e-9); }
        else if (KIND == 1) { a = floor(a * b); }
        else if (KIND == 2) { u = (unsigned)((double)u * 1.00000001) + 1; }   // cvt f64<-u32, mul, cvt u32<-f64
        else if (KIND == 3) { f = fmaf(f, 1.0000001f, 1e-9f); }
        else if (KIND == 4) { u = u * 1664525u + 1013904223u; }                // mul_lo + add
        else if (KIND == 5) { unsigned long long x = ((unsigned long long)u << 32) | u; u = (unsigned)(x / (unsigned long long)(useed | 1)); } // 64-bit div
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + u + f;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
int run(const char *name, int blocks, int n) {
    double *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, blocks * 64 * sizeof(double)));
    CHECK(hipMalloc(&clk, blocks * 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(64), 0, 0, out, clk, n, 1.5, 12345u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double mhz = (double)h[0] / ((double)h[1] / 100.0);   // memrealtime ticks at 100 MHz
    printf("%-28s blocks %4d  n %8d  %8.3f ms  %7.1f ns/iter  %7.1f cyc/iter  shader clock %7.1f MHz\n", name, blocks, n, ms,
           ms * 1e6 / n, (double)h[0] / n, mhz);
    hipFree(out); hipFree(clk);
    return 0;
}

int main() {
    const int n = 2000000;
    for (int blocks : {1, 4, 256, 1024}) {
        run<0>("f64 fma chain", blocks, n);
    }
    run<1>("f64 mul+floor chain", 4, n);
    run<2>("cvt/mul/cvt chain", 4, n);
    run<3>("f32 fma chain", 4, n);
    run<4>("u32 mul+add chain", 4, n);
    run<5>("u64 / u32 divide chain", 4, n / 10);
    return 0;
}
