// Practical ceiling of the f32-operand matrix pipe (v_mfma_f32_16x16x4_f32) on this board: registers only, no memory.
// Each wave issues `iters` x 24 independent-accumulator MFMAs; 1, 2 or 4 waves per SIMD.
//      hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f32_peak.hip -o /tmp/mfma_f32_peak && /tmp/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) float f4_t;

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    f4_t acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, d, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 24 * 2048.0;
            printf("%d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", wgs_per_cu, ms, flop / ms * 1e-9,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 24 * wgs_per_cu));
        }
    }
    return 0;
}
