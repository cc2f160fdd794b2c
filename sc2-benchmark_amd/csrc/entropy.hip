// Entropy-bottleneck element-wise kernels (factorised prior, filters (3,3,3,3)).
//
// Replaces CompressAI EntropyBottleneck.forward (called at sc2bench/models/layer.py:531):
//   permute -> noise / round -> two evaluations of the 5-layer per-channel cumulative-logit MLP ->
//   sigmoid difference -> lower bound 1e-9 -> permute back  (about 40 small torch kernels plus four
//   permute copies upstream) with ONE pass over y: each workgroup walks a slice of one (image,
//   channel) plane of the NCHW latent, so the 59 per-channel parameters are wave-uniform and live in
//   scalar registers; reads and writes are contiguous along H*W.
// HBM-bound: algorithmic bytes per element = 4 (y) [+4 noise] + 4 (y_hat) + 4 (likelihood).
#include "sc2_common.h"

namespace {

constexpr int EB_THREADS = 256;
constexpr int EB_EPT = 4;  // elements per thread
constexpr int EB_TILE = EB_THREADS * EB_EPT;

// cumulative logits L(v) of one channel; P = effective parameters (layout: sc2_bottleneck.h)
__device__ __forceinline__ float eb_logits(float v, const float *__restrict__ P) {
    float h0 = P[0] * v + P[3];
    float h1 = P[1] * v + P[4];
    float h2 = P[2] * v + P[5];
    h0 += P[6] * tanhf(h0);
    h1 += P[7] * tanhf(h1);
    h2 += P[8] * tanhf(h2);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float *Q = P + 9 + 15 * l;
        float g0 = Q[0] * h0 + Q[1] * h1 + Q[2] * h2 + Q[9];
        float g1 = Q[3] * h0 + Q[4] * h1 + Q[5] * h2 + Q[10];
        float g2 = Q[6] * h0 + Q[7] * h1 + Q[8] * h2 + Q[11];
        g0 += Q[12] * tanhf(g0);
        g1 += Q[13] * tanhf(g1);
        g2 += Q[14] * tanhf(g2);
        h0 = g0; h1 = g1; h2 = g2;
    }
    return P[54] * h0 + P[55] * h1 + P[56] * h2 + P[57];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(EB_THREADS) void eb_forward_kernel(
    const float *__restrict__ y, const float *__restrict__ noise, const float *__restrict__ params, int C, int HW,
    int mode, float lik_bound, float *__restrict__ y_hat, uint16_t *__restrict__ y_hat_nhwc,
    float *__restrict__ lik, float *__restrict__ bits_partial) {
    const int plane = blockIdx.x;  // n * C + c
    const int c = plane % C;
    const int n = plane / C;
    const float *P = params + c * SC2_EB_PARAM_STRIDE;
    const float med = P[58];
    const long long base = (long long)plane * HW;
    float bits = 0.f;
#pragma unroll
    for (int e = 0; e < EB_EPT; ++e) {
        const int pix = blockIdx.y * EB_TILE + e * EB_THREADS + threadIdx.x;
        if (pix < HW) {
            const float v = y[base + pix];
            float out;
            if (mode == SC2_EB_NOISE) out = v + noise[base + pix];
            else out = rintf(v - med) + med;
            if (y_hat) y_hat[base + pix] = out;
            if (y_hat_nhwc) y_hat_nhwc[((long long)n * HW + pix) * C + c] = f32_to_bf16_bits(out);
            if (lik || bits_partial) {
                const float lower = eb_logits(out - 0.5f, P);
                const float upper = eb_logits(out + 0.5f, P);
                float l = sigmoidf_(upper) - sigmoidf_(lower);
                l = fmaxf(l, lik_bound);
                if (lik) lik[base + pix] = l;
                bits -= log2f(l);
            }
        }
    }
    if (bits_partial) {
        // workgroup reduction: wave shuffle then LDS
        __shared__ float red[EB_THREADS / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < EB_THREADS / 64; ++w) s += red[w];
            bits_partial[(long long)plane * gridDim.y + blockIdx.y] = s;
        }
    }
}

__global__ __launch_bounds__(256) void eb_symbols_kernel(const float *__restrict__ y, const float *__restrict__ medians,
                                                         int C, int HW, int32_t *__restrict__ symbols) {
    const int plane = blockIdx.x;
    const float med = medians[plane % C];
    const long long base = (long long)plane * HW;
    for (int pix = blockIdx.y * 256 + threadIdx.x; pix < HW; pix += gridDim.y * 256)
        symbols[base + pix] = (int32_t)rintf(y[base + pix] - med);
}

__global__ __launch_bounds__(256) void eb_dequantize_kernel(const int32_t *__restrict__ symbols,
                                                            const float *__restrict__ medians, int C, int HW,
                                                            float *__restrict__ y_hat,
                                                            uint16_t *__restrict__ y_hat_nhwc) {
    const int plane = blockIdx.x;
    const int c = plane % C, n = plane / C;
    const float med = medians[c];
    const long long base = (long long)plane * HW;
    for (int pix = blockIdx.y * 256 + threadIdx.x; pix < HW; pix += gridDim.y * 256) {
        const float v = (float)symbols[base + pix] + med;
        if (y_hat) y_hat[base + pix] = v;
        if (y_hat_nhwc) y_hat_nhwc[((long long)n * HW + pix) * C + c] = f32_to_bf16_bits(v);
    }
}

int plane_grid_x(int HW, int per_block) {
    int g = (HW + per_block - 1) / per_block;
    return g < 1 ? 1 : g;
}

}  // namespace

extern "C" int sc2_eb_bits_partial_len(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return N * C * plane_grid_x(HW, EB_TILE);
}

extern "C" int sc2_eb_forward(const float *y, const float *noise, const float *params, int N, int C, int HW, int mode,
                              float lik_bound, float *y_hat, void *y_hat_bf16_nhwc, float *lik, float *bits_partial,
                              int bits_partial_len, void *stream) {
    SC2_REQUIRE(y && params, SC2_ERR_INVALID_ARG, "eb_forward: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && HW > 0, SC2_ERR_INVALID_ARG, "eb_forward: bad dims N=%d C=%d HW=%d", N, C, HW);
    SC2_REQUIRE(mode == SC2_EB_NOISE || mode == SC2_EB_DEQUANTIZE, SC2_ERR_INVALID_ARG,
                "Invalid quantization mode: \"%d\"", mode);
    if (mode == SC2_EB_NOISE) SC2_REQUIRE(noise, SC2_ERR_INVALID_ARG, "eb_forward: noise mode needs a noise tensor");
    if (bits_partial)
        SC2_REQUIRE(bits_partial_len == sc2_eb_bits_partial_len(N, C, HW), SC2_ERR_INVALID_ARG,
                    "eb_forward: bits_partial_len %d != %d", bits_partial_len, sc2_eb_bits_partial_len(N, C, HW));
    dim3 grid(N * C, plane_grid_x(HW, EB_TILE));
    hipLaunchKernelGGL(eb_forward_kernel, grid, dim3(EB_THREADS), 0, static_cast<hipStream_t>(stream), y, noise,
                       params, C, HW, mode, lik_bound, y_hat, static_cast<uint16_t *>(y_hat_bf16_nhwc), lik,
                       bits_partial);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_eb_symbols(const float *y, const float *medians, int N, int C, int HW, int32_t *symbols,
                              void *stream) {
    SC2_REQUIRE(y && medians && symbols, SC2_ERR_INVALID_ARG, "eb_symbols: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && HW > 0, SC2_ERR_INVALID_ARG, "eb_symbols: bad dims");
    dim3 grid(N * C, plane_grid_x(HW, 1024));
    hipLaunchKernelGGL(eb_symbols_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), y, medians, C, HW,
                       symbols);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_eb_dequantize(const int32_t *symbols, const float *medians, int N, int C, int HW,
                                 float *y_hat_f32_nchw, void *y_hat_bf16_nhwc, void *stream) {
    SC2_REQUIRE(symbols && medians && (y_hat_f32_nchw || y_hat_bf16_nhwc), SC2_ERR_INVALID_ARG,
                "eb_dequantize: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && HW > 0, SC2_ERR_INVALID_ARG, "eb_dequantize: bad dims");
    dim3 grid(N * C, plane_grid_x(HW, 1024));
    hipLaunchKernelGGL(eb_dequantize_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), symbols, medians,
                       C, HW, y_hat_f32_nchw, static_cast<uint16_t *>(y_hat_bf16_nhwc));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
