// dependent-chain latency per instruction kind, 16x unrolled so loop overhead is amortised; one wave per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ void chain(double *out, unsigned long long *clk, int n, double seed, unsigned useed) {
    double a = seed + threadIdx.x, b = 1.0000001;
    unsigned u = useed + threadIdx.x, v = useed * 3 + 1;
    unsigned long long x = ((unsigned long long)useed << 20) + threadIdx.x;
    float f = (float)seed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) a = fma(a, b, 1e-9);
            else if (KIND == 1) a = floor(a * b);
            else if (KIND == 2) a = (double)(unsigned)a * b;           // cvt_u32_f64 + cvt_f64_u32 + mul
            else if (KIND == 3) f = fmaf(f, 1.0000001f, 1e-9f);
            else if (KIND == 4) u = u * 1664525u + 1013904223u;
            else if (KIND == 5) x = (unsigned long long)(unsigned)(x >> 16) * v + (x & 0xFFFF);   // decode-like advance
            else if (KIND == 6) { if ((x >> 47) >= v) { x >>= 32; } x = x * 3 + u; }             // compare/select 64-bit
            else if (KIND == 7) { a = a + ((a >= b) ? -b : 0.0); a = a * 1.5; }                  // cmp + cndmask + add
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + u + f + (double)x;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char *name, int n) {
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 4 * 64 * sizeof(double)); (void)hipMalloc(&clk, 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL(chain<KIND>, dim3(4), dim3(64), 0, 0, out, clk, n, 1.5, 12345u);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s %7.2f cycles per step\n", name, (double)h / (16.0 * n));
    (void)hipFree(out); (void)hipFree(clk);
}
// branch cost: a loop whose body is tiny, with a forward skip branch that is always taken / never taken
__global__ void branches(unsigned *out, unsigned long long *clk, int n, unsigned thr) {
    unsigned u = threadIdx.x, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            u = u * 1664525u + 1013904223u;
            if ((u >> 28) > thr) {            // divergent test; thr=15 -> never, thr=-1 -> always
                acc += out[(u >> 8) & 63];   // something non-trivial so the compiler keeps a real branch
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[64 + threadIdx.x] = acc + u;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    run<0>("f64 fma (dependent)", 100000);
    run<1>("f64 mul + floor", 100000);
    run<2>("cvt_u32_f64 + cvt_f64_u32 + mul_f64", 100000);
    run<3>("f32 fma", 100000);
    run<4>("u32 mul_lo + add", 100000);
    run<5>("u64: (x>>16)*v + (x&0xffff)", 100000);
    run<6>("u64: cmp/shift select + mul3+add", 100000);
    run<7>("f64: cmp+cndmask+add, mul", 100000);
    unsigned *o; unsigned long long *clk; (void)hipMalloc(&o, 1024); (void)hipMemset(o, 0, 1024); (void)hipMalloc(&clk, 8);
    for (unsigned thr : {15u, 7u}) {
        hipLaunchKernelGGL(branches, dim3(1), dim3(64), 0, 0, o, clk, 100000, thr);
        (void)hipDeviceSynchronize();
        unsigned long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        printf("branch test thr=%2u: %7.2f cycles per (mul+add+test[+body])\n", thr, (double)h / 800000.0);
    }
    return 0;
}
