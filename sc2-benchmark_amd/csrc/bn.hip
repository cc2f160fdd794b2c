// BatchNorm2d with BATCH statistics on a bf16 NHWC map, forward and backward, with the ReLU and the residual add of a ResNet
// Bottleneck block folded in (gfx950).  Stage 2 of the Entropic-Student recipe fine-tunes the student's layer2 .. layer4 with their
// norm layers in training mode (configs/ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_
// resnet50.yaml:231-295; the blocks: torchvision Bottleneck, sc2bench/models/backbone.py:235-254 runs them).  As torch / MIOpen ops
// a block's three norm layers, two ReLUs and the add are ~16 passes over its maps forward and as many backward (profiles/
// r05q_train2_kernel_stats.csv: 9.9 ms of norm kernels + 3.7 ms of ReLU / add kernels per 256-image step); here
//     forward   y = relu?( (x - mean) * rstd * gamma + beta  (+ residual) )         : one reduction pass + one apply pass
//     backward  dz = dy * (y > 0)?;  dgamma = sum dz * xhat;  dbeta = sum dz;
//               dx = gamma * rstd * (dz - dbeta / M - xhat * dgamma / M)  (+ dz handed on to the residual's producer)
//                                                                                    : one reduction pass + one apply pass
// Every pass is a stream over [M = N H W][C] bf16 rows, 16 bytes per thread and load, f32 arithmetic.  The per-channel algebra
// between the passes (mean, variance, the running statistics, the three coefficients of the backward apply pass) runs in a
// small kernel, so a layer is three launches each way and no host round trip.
// Statistics as nn.BatchNorm2d computes them: biased variance for the normalisation, unbiased for running_var, momentum update
// of both; E[x^2] - mean^2 in f32 over bf16 inputs (conv outputs, |mean| of the order of the deviation: no cancellation to speak of).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sc2_common.h"

namespace {

constexpr int BN_THREADS = 512, BN_UNROLL = 2;     // (the LDS reduction holds 17 floats per thread: 35 KB)

__device__ __forceinline__ void bn_unpack8(const uint4 v, float (&a)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[2 * q] = __builtin_bit_cast(float, w[q] << 16);
        a[2 * q + 1] = __builtin_bit_cast(float, w[q] & 0xFFFF0000u);
    }
}
typedef float bn_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bn_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 bn_pack8(const float (&a)[8]) {   // round-to-nearest-even, as torch's casts
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(bn_f32x2_t{a[2 * q], a[2 * q + 1]}, bn_bf16x2_t));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Partial sums of workgroup b go to out[b][0 .. 2C) (no atomics, no memset: 512 workgroups adding into the same 2 C addresses
// serialise in L2 -- ~0.15 us per add and address -- and a 2 048-channel layer's reduction took 80 us for 51 MB); the finalize
// kernels add the partials up.
// MODE 0: out[b][0..C) = sum x, out[b][C..2C) = sum x^2 over the workgroup's rows.
// MODE 1: dz = y ? dy * (y > 0) : dy;  out[b][0..C) = sum dz, out[b][C..2C) = sum dz * (x - mean) * rstd.
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                               const uint16_t *__restrict__ y, const float *__restrict__ mean,
                                                               const float *__restrict__ rstd, long long M, int C, float *__restrict__ out) {
    const int cpr = C >> 3;
    const int ppi = BN_THREADS / cpr;                // rows a workgroup covers per iteration
    const int cc = threadIdx.x % cpr, pl = threadIdx.x / cpr;
    float s0[8], s1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) s0[t] = s1[t] = 0.f;
    if (pl < ppi) {
        float mu[8], rs[8];
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                mu[t] = mean[cc * 8 + t];
                rs[t] = rstd[cc * 8 + t];
            }
        }
        const long long stride = (long long)gridDim.x * ppi;
        for (long long m = (long long)blockIdx.x * ppi + pl; m < M; m += BN_UNROLL * stride) {
            uint4 rx[BN_UNROLL], rd[BN_UNROLL], ry[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long mu_ = m + u * stride;
                const long long off = (mu_ < M ? mu_ : m) * C + cc * 8;
                rx[u] = *reinterpret_cast<const uint4 *>(x + off);
                if (MODE == 1) {
                    rd[u] = *reinterpret_cast<const uint4 *>(dy + off);
                    ry[u] = y ? *reinterpret_cast<const uint4 *>(y + off) : make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
                }
            }
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const bool live = m + u * stride < M;
                float a[8];
                bn_unpack8(rx[u], a);
                if (MODE == 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float v = live ? a[t] : 0.f;
                        s0[t] += v;
                        s1[t] += v * v;
                    }
                } else {
                    float d[8], o[8];
                    bn_unpack8(rd[u], d);
                    bn_unpack8(ry[u], o);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float dz = (live && o[t] > 0.f) ? d[t] : 0.f;
                        s0[t] += dz;
                        s1[t] += dz * ((a[t] - mu[t]) * rs[t]);
                    }
                }
            }
        }
    }
    __shared__ float red[BN_THREADS][17];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        red[threadIdx.x][t] = s0[t];
        red[threadIdx.x][8 + t] = s1[t];
    }
    __syncthreads();
    if (threadIdx.x < cpr) {
        float tot[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) tot[t] = 0.f;
        for (int q = threadIdx.x; q < ppi * cpr; q += cpr)
#pragma unroll
            for (int t = 0; t < 16; ++t) tot[t] += red[q][t];
        float *o = out + (long long)blockIdx.x * 2 * C;
        *reinterpret_cast<float4 *>(o + threadIdx.x * 8) = make_float4(tot[0], tot[1], tot[2], tot[3]);
        *reinterpret_cast<float4 *>(o + threadIdx.x * 8 + 4) = make_float4(tot[4], tot[5], tot[6], tot[7]);
        *reinterpret_cast<float4 *>(o + C + threadIdx.x * 8) = make_float4(tot[8], tot[9], tot[10], tot[11]);
        *reinterpret_cast<float4 *>(o + C + threadIdx.x * 8 + 4) = make_float4(tot[12], tot[13], tot[14], tot[15]);
    }
}

// Adds up the per-workgroup partial rows for 32 channels per workgroup: thread (channel c = tid & 31, slice = tid >> 5) takes
// partial rows slice, slice + 8, ... four at a time (independent loads: a serial walk over 512 rows cost ~100 us of L2 latency
// per layer), the eight slices meet in LDS.  -> (sum of column c, sum of column C + c) in the threads of slice 0.
__device__ __forceinline__ void bn_sum_partials(const float *__restrict__ part, int n_part, int C, int c, int slice, bool live, float &s0,
                                                float &s1) {
    __shared__ float red[2][8][32];
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        int b = slice;
        for (; b + 24 < n_part; b += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] += part[(long long)(b + 8 * u) * 2 * C + c];
                a1[u] += part[(long long)(b + 8 * u) * 2 * C + C + c];
            }
        }
        for (; b < n_part; b += 8) {
            a0[0] += part[(long long)b * 2 * C + c];
            a1[0] += part[(long long)b * 2 * C + C + c];
        }
    }
    red[0][slice][threadIdx.x & 31] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
    red[1][slice][threadIdx.x & 31] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    __syncthreads();
    s0 = s1 = 0.f;
    if (slice == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s0 += red[0][q][threadIdx.x & 31];
            s1 += red[1][q][threadIdx.x & 31];
        }
    }
}

// partial sums -> mean, rstd (saved for the backward), scale / shift of the apply pass, running statistics (momentum update);
// grid = C / 32 workgroups of 256 threads
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float *__restrict__ part, int n_part, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ running_mean,
                                                              float *__restrict__ running_var, float momentum, float eps, long long M, int C,
                                                              float *__restrict__ save_mean, float *__restrict__ save_rstd,
                                                              float *__restrict__ scale, float *__restrict__ shift) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), slice = threadIdx.x >> 5;
    float su, sq;
    bn_sum_partials(part, n_part, C, c, slice, c < C, su, sq);
    if (slice == 0 && c < C) {
        const float inv_m = 1.0f / (float)M;
        const float mu = su * inv_m;
        float var = sq * inv_m - mu * mu;
        var = var > 0.f ? var : 0.f;
        const float rs = 1.0f / __builtin_sqrtf(var + eps);
        save_mean[c] = mu;
        save_rstd[c] = rs;
        const float sc = gamma[c] * rs;
        scale[c] = sc;
        shift[c] = beta[c] - mu * sc;
        if (running_mean) {
            const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

// y = relu?(x * scale + shift (+ residual))
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ residual,
                                                       const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                       long long chunks, int cpr, uint16_t *__restrict__ y) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += stride) {
        const int c0 = (int)(q % cpr) * 8;
        float a[8], r[8], o[8];
        bn_unpack8(reinterpret_cast<const uint4 *>(x)[q], a);
        if (residual) bn_unpack8(reinterpret_cast<const uint4 *>(residual)[q], r);
        const float4 sc0 = *reinterpret_cast<const float4 *>(scale + c0), sc1 = *reinterpret_cast<const float4 *>(scale + c0 + 4);
        const float4 sh0 = *reinterpret_cast<const float4 *>(shift + c0), sh1 = *reinterpret_cast<const float4 *>(shift + c0 + 4);
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float v = a[t] * sc[t] + sh[t];
            if (residual) v += r[t];
            o[t] = (relu && !(v > 0.f)) ? 0.f : v;
        }
        reinterpret_cast<uint4 *>(y)[q] = bn_pack8(o);
    }
}

// sums [2C] = (sum dz, sum dz xhat) -> dbeta, dgamma and the coefficients of dx = ca dz + cb x + cc
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ part, int n_part, const float *__restrict__ gamma,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd, long long M,
                                                              int C, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                              float *__restrict__ ca, float *__restrict__ cb, float *__restrict__ cc) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), slice = threadIdx.x >> 5;
    float db, dg;
    bn_sum_partials(part, n_part, C, c, slice, c < C, db, dg);
    if (slice == 0 && c < C) {
        const float inv_m = 1.0f / (float)M;
        dbeta[c] = db;
        dgamma[c] = dg;
        const float g_rs = gamma[c] * rstd[c];
        ca[c] = g_rs;
        const float b = -g_rs * rstd[c] * dg * inv_m;      // coefficient of x:  - gamma rstd * (rstd * dgamma / M)
        cb[c] = b;
        cc[c] = -g_rs * db * inv_m - b * mean[c];
    }
}

// dz = y ? dy * (y > 0) : dy;  dx = ca dz + cb x + cc;  dz_out (the gradient of the residual operand), when asked for
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                           const uint16_t *__restrict__ y, const float *__restrict__ ca,
                                                           const float *__restrict__ cb, const float *__restrict__ cc, long long chunks, int cpr,
                                                           uint16_t *__restrict__ dx, uint16_t *__restrict__ dz_out) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += stride) {
        const int c0 = (int)(q % cpr) * 8;
        float d[8], a[8], o[8], g[8], z[8];
        bn_unpack8(reinterpret_cast<const uint4 *>(dy)[q], d);
        bn_unpack8(reinterpret_cast<const uint4 *>(x)[q], a);
        if (y) bn_unpack8(reinterpret_cast<const uint4 *>(y)[q], o);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float dz = (!y || o[t] > 0.f) ? d[t] : 0.f;
            z[t] = dz;
            g[t] = ca[c0 + t] * dz + cb[c0 + t] * a[t] + cc[c0 + t];
        }
        reinterpret_cast<uint4 *>(dx)[q] = bn_pack8(g);
        if (dz_out) reinterpret_cast<uint4 *>(dz_out)[q] = bn_pack8(z);
    }
}

inline unsigned bn_reduce_blocks(long long M, int C) {
    const int ppi = BN_THREADS / (C / 8);
    long long blocks = (M + (long long)ppi * BN_UNROLL * 4 - 1) / ((long long)ppi * BN_UNROLL * 4);   // at least four trips per workgroup
    if (blocks > 512) blocks = 512;      // (two workgroups per CU; the finalize kernel adds up that many partial rows)
    return (unsigned)(blocks < 1 ? 1 : blocks);
}
inline unsigned bn_stream_blocks(long long chunks) {
    long long blocks = (chunks + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    return (unsigned)blocks;
}

}  // namespace

extern "C" long long sc2_bn_ws_floats(long long M, int C) {   // scratch of either direction: partial sums + coefficient rows
    if (M <= 0 || C <= 0 || C % 8 != 0 || C / 8 > 256) return 0;
    return (long long)bn_reduce_blocks(M, C) * 2 * C + 3 * C;
}

extern "C" int sc2_bn_train_fwd(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, float momentum, float eps, int relu, void *y, float *save_mean, float *save_rstd,
                                float *ws, long long M, int C, void *stream) {
    SC2_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && ws, SC2_ERR_INVALID_ARG, "bn_train_fwd: null argument");
    SC2_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SC2_ERR_INVALID_ARG, "bn_train_fwd: running_mean and running_var go together");
    SC2_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256, SC2_ERR_INVALID_ARG, "bn_train_fwd: bad dims M=%lld C=%d (C %% 8 == 0, C <= 2048)", M, C);
    // one value per channel has no variance: nn.BatchNorm2d refuses it in training mode ("Expected more than 1 value per channel"), so does this
    SC2_REQUIRE(M > 1, SC2_ERR_INVALID_ARG, "bn_train_fwd: expected more than 1 value per channel when training (M=%lld)", M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned nb = bn_reduce_blocks(M, C);
    float *coef = ws + (size_t)nb * 2 * C;          // scale, shift behind the partial sums
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nb), dim3(BN_THREADS), 0, s, static_cast<const uint16_t *>(x), nullptr, nullptr, nullptr,
                       nullptr, M, C, ws);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, ws, (int)nb, gamma, beta, running_mean, running_var,
                       momentum, eps, M, C, save_mean, save_rstd, coef, coef + C);
    const long long chunks = M * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_stream_blocks(chunks)), dim3(256), 0, s, static_cast<const uint16_t *>(x),
                       static_cast<const uint16_t *>(residual), coef, coef + C, relu ? 1 : 0, chunks, C / 8, static_cast<uint16_t *>(y));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_bn_train_bwd(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean,
                                const float *save_rstd, void *dx, void *dz, float *dgamma, float *dbeta, float *ws, long long M, int C,
                                void *stream) {
    SC2_REQUIRE(dy && x && gamma && save_mean && save_rstd && dx && dgamma && dbeta && ws, SC2_ERR_INVALID_ARG, "bn_train_bwd: null argument");
    SC2_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256, SC2_ERR_INVALID_ARG, "bn_train_bwd: bad dims M=%lld C=%d (C %% 8 == 0, C <= 2048)", M, C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned nb = bn_reduce_blocks(M, C);
    float *coef = ws + (size_t)nb * 2 * C;          // the three coefficient rows behind the partial sums
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nb), dim3(BN_THREADS), 0, s, static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(dy),
                       static_cast<const uint16_t *>(y), save_mean, save_rstd, M, C, ws);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, ws, (int)nb, gamma, save_mean, save_rstd, M, C, dgamma,
                       dbeta, coef, coef + C, coef + 2 * C);
    const long long chunks = M * (C / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_stream_blocks(chunks)), dim3(256), 0, s, static_cast<const uint16_t *>(dy),
                       static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(y), coef, coef + C, coef + 2 * C, chunks, C / 8,
                       static_cast<uint16_t *>(dx), static_cast<uint16_t *>(dz));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
