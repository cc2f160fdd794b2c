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

// Backward of eb_forward_kernel.  Per element the 5-layer cumulative-logit MLP is re-evaluated at y_hat -/+ 1/2
// (nothing but y_hat is saved by the forward) and differentiated by hand; the 58 per-channel parameter gradients are
// accumulated in registers over the thread's elements, reduced over the workgroup (wave shuffles, then LDS) and written
// as one partial row per workgroup; the caller sums the partial rows of a channel.
//   g_lik passes the likelihood lower bound by CompressAI's LowerBound rule (x >= bound or the gradient pushes x up).
// (the forward of one evaluation with everything its backward needs: 12 tanh values, the three hidden vectors)
struct EbEval {
    float v, t0[3], h0[3], tl[3][3], hl[3][3];
};

__device__ __forceinline__ float eb_logits_keep(float v, const float *__restrict__ P, EbEval &e) {
    e.v = v;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g0 = P[k] * v + P[3 + k];
        e.t0[k] = tanhf(g0);
        e.h0[k] = g0 + P[6 + k] * e.t0[k];
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float *Q = P + 9 + 15 * l;
        const float *hin = l == 0 ? e.h0 : e.hl[l - 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gl = Q[3 * k] * hin[0] + Q[3 * k + 1] * hin[1] + Q[3 * k + 2] * hin[2] + Q[9 + k];
            e.tl[l][k] = tanhf(gl);
            e.hl[l][k] = gl + Q[12 + k] * e.tl[l][k];
        }
    }
    return P[54] * e.hl[2][0] + P[55] * e.hl[2][1] + P[56] * e.hl[2][2] + P[57];
}

// backward of one kept evaluation: parameter gradients into G, returns d out / d v
__device__ __forceinline__ float eb_logits_bwd_kept(const EbEval &e, const float *__restrict__ P, float d_out, float *__restrict__ G) {
    float dh[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[54 + j] += d_out * e.hl[2][j];
        dh[j] = d_out * P[54 + j];
    }
    G[57] += d_out;
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        const float *Q = P + 9 + 15 * l;
        float *GQ = G + 9 + 15 * l;
        const float *hin = l == 0 ? e.h0 : e.hl[l - 1];
        float dg[3], dhin[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            GQ[12 + k] += dh[k] * e.tl[l][k];
            dg[k] = dh[k] * (1.0f + Q[12 + k] * (1.0f - e.tl[l][k] * e.tl[l][k]));
            GQ[9 + k] += dg[k];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                GQ[3 * k + j] += dg[k] * hin[j];
                dhin[j] += Q[3 * k + j] * dg[k];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) dh[j] = dhin[j];
    }
    float dv = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        G[6 + k] += dh[k] * e.t0[k];
        const float dg = dh[k] * (1.0f + P[6 + k] * (1.0f - e.t0[k] * e.t0[k]));
        G[3 + k] += dg;
        G[k] += dg * e.v;
        dv += P[k] * dg;
    }
    return dv;
}

__global__ __launch_bounds__(EB_THREADS) void eb_backward_kernel(
    const float *__restrict__ y, const float *__restrict__ noise, const float *__restrict__ params, int C, int HW,
    int mode, float lik_bound, const float *__restrict__ g_yhat, const float *__restrict__ g_lik,
    float *__restrict__ g_y, float *__restrict__ g_partial, int planes_per_wg, int rows_per_plane) {
    // One workgroup walks `planes_per_wg` WHOLE planes of one channel (images n0 .. n0 + planes_per_wg - 1): the 59
    // parameter-gradient sums per thread end in 354 cross-lane shuffles + an LDS round, which with four elements per thread cost as
    // much as the elements themselves (round 4: 0.85 ms per 256 x 24 x 55 x 55 launch with a workgroup per 1 024 elements, 0.66
    // with one per plane).  The partial-sum rows the caller adds up keep their layout: this workgroup's sums go to the first row of
    // its first plane, zeros to the other rows of its planes.
    const int c = blockIdx.x % C, n0 = (blockIdx.x / C) * planes_per_wg;
    const float *P = params + c * SC2_EB_PARAM_STRIDE;
    const float med = P[58];
    float G[SC2_EB_PARAM_STRIDE];
#pragma unroll
    for (int k = 0; k < SC2_EB_PARAM_STRIDE; ++k) G[k] = 0.f;
    float g_med = 0.f;
#pragma unroll 1
    for (int e = threadIdx.x; e < planes_per_wg * HW; e += EB_THREADS) {
        const int pi = e / HW, pix = e - pi * HW;
        const long long base = ((long long)(n0 + pi) * C + c) * HW;
        {
            const float v = y[base + pix];
            const float out = mode == SC2_EB_NOISE ? v + noise[base + pix] : rintf(v - med) + med;
            float d_total = g_yhat ? g_yhat[base + pix] : 0.f;
            if (g_lik) {
                const float gl = g_lik[base + pix];
                // (both evaluations are kept: their backward passes used to recompute them -- 24 of the element's 48 tanhf)
                EbEval el, eu;
                const float lower = eb_logits_keep(out - 0.5f, P, el);
                const float upper = eb_logits_keep(out + 0.5f, P, eu);
                const float su = sigmoidf_(upper), sl = sigmoidf_(lower);
                const float raw = su - sl;
                const float g_raw = (raw >= lik_bound || gl < 0.f) ? gl : 0.f;
                if (g_raw != 0.f) {
                    d_total += eb_logits_bwd_kept(eu, P, g_raw * su * (1.0f - su), G);
                    d_total += eb_logits_bwd_kept(el, P, -g_raw * sl * (1.0f - sl), G);
                }
            }
            // noise mode: y_hat = y + u -> dy = d y_hat ; dequantize mode: y_hat = round(y - m) + m -> dy = 0, dm = d y_hat
            if (mode == SC2_EB_NOISE) {
                g_y[base + pix] = d_total;
            } else {
                g_y[base + pix] = 0.f;
                g_med += d_total;
            }
        }
    }
    G[58] = g_med;
    G[63] = 0.f;
    // workgroup reduction of the 59 sums
    __shared__ float red[EB_THREADS / 64][SC2_EB_PARAM_STRIDE];
#pragma unroll
    for (int k = 0; k < 59; ++k) {
        float s = G[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < planes_per_wg * rows_per_plane * SC2_EB_PARAM_STRIDE; q += EB_THREADS) {
        const int pi = q / (rows_per_plane * SC2_EB_PARAM_STRIDE), rem = q - pi * (rows_per_plane * SC2_EB_PARAM_STRIDE);
        float s = 0.f;
        if (q < 59) {                        // (row 0 of the first plane: q = parameter index)
#pragma unroll
            for (int w = 0; w < EB_THREADS / 64; ++w) s += red[w][q];
        }
        g_partial[((long long)(n0 + pi) * C + c) * rows_per_plane * SC2_EB_PARAM_STRIDE + rem] = s;
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

// Same, through an LDS tile so that the bf16 NHWC copy leaves as whole 16-byte runs: a block owns 256 pixels of one image
// for ALL channels (symbols read per channel plane, coalesced), C even.  The plane-per-block form above scatters two bytes
// per lane at a stride of 2 C bytes: 328 MB of HBM writes for a 37 MB tensor (PMC WRITE_SIZE, profiles/r01f_pmc_traffic.txt).
constexpr int DQ_TILE = 256;
__global__ __launch_bounds__(256) void eb_dequantize_tile_kernel(const int32_t *__restrict__ symbols,
                                                                 const float *__restrict__ medians, int C, int HW,
                                                                 float *__restrict__ y_hat,
                                                                 uint16_t *__restrict__ y_hat_nhwc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dq_smem[];
    uint16_t *tile = reinterpret_cast<uint16_t *>(dq_smem);     // [DQ_TILE][C]
    const int n = blockIdx.y;
    const int pix0 = blockIdx.x * DQ_TILE;
    const int npix = HW - pix0 < DQ_TILE ? HW - pix0 : DQ_TILE;
    const int t = threadIdx.x;
    for (int c = 0; c < C; ++c) {
        const long long base = ((long long)n * C + c) * HW + pix0;
        if (t < npix) {
            const float v = (float)symbols[base + t] + medians[c];
            if (y_hat) y_hat[base + t] = v;
            tile[t * C + c] = f32_to_bf16_bits(v);
        }
    }
    __syncthreads();
    const int bytes = npix * C * 2;                              // contiguous in the NHWC output
    unsigned char *dst = reinterpret_cast<unsigned char *>(y_hat_nhwc + ((long long)n * HW + pix0) * C);
    // (pix0 * C * 2 is a multiple of 16 for DQ_TILE = 256 and even C; the image base is when HW * C * 2 is)
    if (((uintptr_t)dst & 15) == 0) {
        for (int q = t; q < bytes / 16; q += 256) reinterpret_cast<uint4 *>(dst)[q] = reinterpret_cast<const uint4 *>(dq_smem)[q];
        for (int q = (bytes / 16) * 16 + t; q < bytes; q += 256) dst[q] = dq_smem[q];
    } else {
        for (int q = t; q < bytes / 4; q += 256) reinterpret_cast<uint32_t *>(dst)[q] = reinterpret_cast<const uint32_t *>(dq_smem)[q];
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

extern "C" int sc2_eb_backward(const float *y, const float *noise, const float *params, int N, int C, int HW, int mode,
                               float lik_bound, const float *g_yhat, const float *g_lik, float *g_y,
                               float *g_params_partial, int n_partial, void *stream) {
    SC2_REQUIRE(y && params && g_y && g_params_partial, SC2_ERR_INVALID_ARG, "eb_backward: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && HW > 0, SC2_ERR_INVALID_ARG, "eb_backward: bad dims N=%d C=%d HW=%d", N, C, HW);
    SC2_REQUIRE(mode == SC2_EB_NOISE || mode == SC2_EB_DEQUANTIZE, SC2_ERR_INVALID_ARG,
                "Invalid quantization mode: \"%d\"", mode);
    if (mode == SC2_EB_NOISE) SC2_REQUIRE(noise, SC2_ERR_INVALID_ARG, "eb_backward: noise mode needs the noise tensor");
    SC2_REQUIRE(n_partial == sc2_eb_bits_partial_len(N, C, HW), SC2_ERR_INVALID_ARG,
                "eb_backward: n_partial %d != %d", n_partial, sc2_eb_bits_partial_len(N, C, HW));
    int ppw = 8;                             // planes (images) per workgroup: a divisor of N that leaves >= 512 workgroups
    while (ppw > 1 && (N % ppw != 0 || (long long)(N / ppw) * C < 512)) ppw >>= 1;
    dim3 grid((N / ppw) * C, 1);
    hipLaunchKernelGGL(eb_backward_kernel, grid, dim3(EB_THREADS), 0, static_cast<hipStream_t>(stream), y, noise,
                       params, C, HW, mode, lik_bound, g_yhat, g_lik, g_y, g_params_partial, ppw, plane_grid_x(HW, EB_TILE));
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
    if (y_hat_bf16_nhwc && C % 2 == 0 && C <= 128 && N <= 65535) {   // tile form: coalesced NHWC stores
        dim3 grid(plane_grid_x(HW, DQ_TILE), N);
        hipLaunchKernelGGL(eb_dequantize_tile_kernel, grid, dim3(256), (size_t)DQ_TILE * C * 2, static_cast<hipStream_t>(stream),
                           symbols, medians, C, HW, y_hat_f32_nchw, static_cast<uint16_t *>(y_hat_bf16_nhwc));
        SC2_CHECK_LAUNCH();
        return SC2_OK;
    }
    dim3 grid(N * C, plane_grid_x(HW, 1024));
    hipLaunchKernelGGL(eb_dequantize_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), symbols, medians,
                       C, HW, y_hat_f32_nchw, static_cast<uint16_t *>(y_hat_bf16_nhwc));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
