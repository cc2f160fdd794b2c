// Layout conversion kernels (HBM-bound, coalesced on the wide side).
//   f32 NCHW (what reference callers hand to nn.Conv2d, sc2bench/models/layer.py:475)
//   <-> bf16 NHWC (the layout every implicit-GEMM kernel of this library consumes/produces).
#include "sc2_common.h"

namespace {

// one thread = one pixel x VEC channels.  Lanes run along pixels, so the per-plane f32 reads are
// contiguous; each lane writes one VEC*2-byte vector.
template <int VEC>
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float *__restrict__ x,
                                                                    uint16_t *__restrict__ y, int C, int HW, int Cpad,
                                                                    long long total_pix) {
    const int cg = blockIdx.y;  // channel group
    const int c0 = cg * VEC;
    for (long long gp = (long long)blockIdx.x * 256 + threadIdx.x; gp < total_pix; gp += (long long)gridDim.x * 256) {
        const long long n = gp / HW;
        const int pix = (int)(gp - n * HW);
        uint16_t v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = c0 + j;
            v[j] = c < C ? f32_to_bf16_bits(x[(n * C + c) * HW + pix]) : (uint16_t)0;
        }
        uint16_t *dst = y + gp * Cpad + c0;
        if (VEC == 8) {
            uint4 o;
            o.x = v[0] | ((uint32_t)v[1] << 16);
            o.y = v[2] | ((uint32_t)v[3] << 16);
            o.z = v[4] | ((uint32_t)v[5] << 16);
            o.w = v[6] | ((uint32_t)v[7] << 16);
            *reinterpret_cast<uint4 *>(dst) = o;
        } else {
            uint2 o;
            o.x = v[0] | ((uint32_t)v[1] << 16);
            o.y = v[2] | ((uint32_t)v[3] << 16);
            *reinterpret_cast<uint2 *>(dst) = o;
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const uint16_t *__restrict__ x,
                                                                    float *__restrict__ y, int C, int HW,
                                                                    long long total_pix) {
    const int c0 = blockIdx.y * 8;
    for (long long gp = (long long)blockIdx.x * 256 + threadIdx.x; gp < total_pix; gp += (long long)gridDim.x * 256) {
        const long long n = gp / HW;
        const int pix = (int)(gp - n * HW);
        const uint4 r = *reinterpret_cast<const uint4 *>(x + gp * C + c0);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            y[(n * C + c0 + 2 * t) * HW + pix] = __builtin_bit_cast(float, w[t] << 16);
            y[(n * C + c0 + 2 * t + 1) * HW + pix] = __builtin_bit_cast(float, w[t] & 0xFFFF0000u);
        }
    }
}

}  // namespace

extern "C" int sc2_nchw_f32_to_nhwc_bf16(const float *x, void *y, int N, int C, int H, int W, int Cpad, void *stream) {
    SC2_REQUIRE(x && y, SC2_ERR_INVALID_ARG, "nchw_to_nhwc: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 4 == 0, SC2_ERR_INVALID_ARG,
                "nchw_to_nhwc: bad dims N=%d C=%d H=%d W=%d Cpad=%d", N, C, H, W, Cpad);
    const long long total = (long long)N * H * W;
    const int gx = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cpad % 8 == 0)
        hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<8>, dim3(gx, Cpad / 8), dim3(256), 0, s, x,
                           static_cast<uint16_t *>(y), C, H * W, Cpad, total);
    else
        hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel<4>, dim3(gx, Cpad / 4), dim3(256), 0, s, x,
                           static_cast<uint16_t *>(y), C, H * W, Cpad, total);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

namespace {
// AdaptiveAvgPool2d((1, 1)) + flatten on a bf16 NHWC feature map: one thread per 8 channels of one image walks the HW
// pixels (each load a 16-byte channel run, a wave reads 1 KB contiguous), f32 accumulation, mean rounded once.
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const uint16_t *__restrict__ x, float *__restrict__ y_f32,
                                                           uint16_t *__restrict__ y_bf16, int HW, int C) {
    // 64 channel octets x 4 pixel groups per workgroup: four times the loads in flight of one thread per octet (the 7 x 7 x 2048
    // map of 256 images: 0.018 -> 0.011 ms); partial sums meet in LDS, in a fixed order
    __shared__ float part[3][64][8];
    const int n = blockIdx.y;
    const int oc = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c8 = blockIdx.x * 64 + oc;
    const bool on = c8 * 8 < C;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
        const uint4 *src = reinterpret_cast<const uint4 *>(x + (long long)n * HW * C) + c8;
        for (int p = pg; p < HW; p += 4) {
            const uint4 v = src[(long long)p * (C / 8)];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += __builtin_bit_cast(float, w[k] << 16);
                acc[2 * k + 1] += __builtin_bit_cast(float, w[k] & 0xFFFF0000u);
            }
        }
    }
    if (pg > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) part[pg - 1][oc][k] = acc[k];
    }
    __syncthreads();
    if (pg != 0 || !on) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += part[g][oc][k];
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= inv;
    if (y_f32) {
        float4 *o = reinterpret_cast<float4 *>(y_f32 + (long long)n * C + c8 * 8);
        o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (y_bf16) {
        uint4 o;
        o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
        o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
        *reinterpret_cast<uint4 *>(y_bf16 + (long long)n * C + c8 * 8) = o;
    }
}

// Classifier on the pooled features (torchvision ResNet.fc behind AdaptiveAvgPool2d + flatten, sc2bench/models/backbone.py:
// 247-253): out[m][n] = sum_k a[m][k] w[n][k] + bias[n], M = batch (256), K = 2048, N = 1000.  As a 1x1 conv on the tile kernel
// this is 16 workgroups walking K = 2048 serially (0.052 ms); here a workgroup owns 16 outputs x 128 images, its four waves take
// a quarter of K each (operands straight from L2 into registers, no LDS in the loop) and meet once in LDS: 126 workgroups.
constexpr int FC_RT = 8;   // 16-image row tiles per workgroup

__global__ __launch_bounds__(256) void fc_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ w_frag,
                                                 const float *__restrict__ bias, float *__restrict__ out, int M, int K, int Npad) {
    __shared__ __attribute__((aligned(16))) float red[3][FC_RT][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int jt = blockIdx.x, m0 = blockIdx.y * (FC_RT * 16);
    const int KS = K >> 5, ks0 = wave * (KS >> 2), ks1 = ks0 + (KS >> 2);
    const uint4 *wp = reinterpret_cast<const uint4 *>(w_frag) + ((long long)jt * KS) * 64 + lane;
    const uint4 *ap[FC_RT];
#pragma unroll
    for (int i = 0; i < FC_RT; ++i) {
        int m = m0 + i * 16 + frow;
        m = m < M ? m : M - 1;   // (rows past the batch: any valid row, never stored)
        ap[i] = reinterpret_cast<const uint4 *>(a + (long long)m * K + fq * 8);
    }
    f32x4_t acc[FC_RT];
#pragma unroll
    for (int i = 0; i < FC_RT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int ks = ks0; ks < ks1; ++ks) {
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wp[(long long)ks * 64]);
        uint4 av[FC_RT];
#pragma unroll
        for (int i = 0; i < FC_RT; ++i) av[i] = ap[i][ks * 4];
#pragma unroll
        for (int i = 0; i < FC_RT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, av[i]), acc[i], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < FC_RT; ++i) *reinterpret_cast<float4 *>(red[wave - 1][i][lane]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    __syncthreads();
    if (wave != 0) return;
    // lane (frow, fq) holds outputs n = 16 jt + 4 fq + [0, 4) of image m0 + 16 i + frow
    const float4 b = *reinterpret_cast<const float4 *>(bias + jt * 16 + 4 * fq);
#pragma unroll
    for (int i = 0; i < FC_RT; ++i) {
        float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 r = *reinterpret_cast<const float4 *>(red[g][i][lane]);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        const int m = m0 + i * 16 + frow;
        if (m < M) *reinterpret_cast<float4 *>(out + (long long)m * Npad + jt * 16 + 4 * fq) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    }
}
}  // namespace

namespace {
// MaxPool2d on a bf16 NHWC map (torchvision ResNet.maxpool behind the stem: k3 s2 p1 on [N, 112, 112, 64]; the teacher's forward
// of the training step and the input-compression classifier run it).  One thread = one output pixel x 8 channels: KH x KW 16-byte
// loads (neighbouring windows overlap in L1 / L2; HBM sees the map once), one 16-byte store.  The update rule is torch's --
// `if (v > m || isnan(v)) m = v` over the window in row-major order, from -inf -- so the result is bit-identical to
// nn.functional.max_pool2d including NaN propagation and the sign of zero.
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int H, int W, int C8,
                                                           int OH, int OW, int KH, int KW, int SH, int SW, int PH, int PW, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    long long px = t / C8;
    const int ow = (int)(px % OW);
    px /= OW;
    const int oh = (int)(px % OH);
    const long long n = px / OH;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
    const uint4 *xi = reinterpret_cast<const uint4 *>(x) + n * H * W * C8 + c8;
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh * SH - PH + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = ow * SW - PW + kw;
            if ((unsigned)iw >= (unsigned)W) continue;
            const uint4 v = xi[((long long)ih * W + iw) * C8];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __builtin_bit_cast(float, w4[q] << 16), hi = __builtin_bit_cast(float, w4[q] & 0xFFFF0000u);
                if (lo > m[2 * q] || lo != lo) m[2 * q] = lo;
                if (hi > m[2 * q + 1] || hi != hi) m[2 * q + 1] = hi;
            }
        }
    }
    uint4 o;
    uint32_t *ow4 = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        ow4[q] = (__builtin_bit_cast(uint32_t, m[2 * q]) >> 16) | (__builtin_bit_cast(uint32_t, m[2 * q + 1]) & 0xFFFF0000u);
    reinterpret_cast<uint4 *>(y)[t] = o;
}
}  // namespace

extern "C" int sc2_maxpool_nhwc(const void *x, void *y, int N, int H, int W, int C, int KH, int KW, int stride_h, int stride_w, int pad_h,
                                int pad_w, void *stream) {
    SC2_REQUIRE(x && y, SC2_ERR_INVALID_ARG, "maxpool_nhwc: null argument");
    SC2_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, SC2_ERR_INVALID_ARG, "maxpool_nhwc: bad dims N=%d H=%d W=%d C=%d (C %% 8 == 0)", N,
                H, W, C);
    SC2_REQUIRE(KH > 0 && KW > 0 && stride_h > 0 && stride_w > 0 && pad_h >= 0 && pad_w >= 0 && 2 * pad_h <= KH && 2 * pad_w <= KW,
                SC2_ERR_INVALID_ARG, "maxpool_nhwc: bad window (padding at most half the kernel, as nn.MaxPool2d requires)");
    const int OH = (H + 2 * pad_h - KH) / stride_h + 1, OW = (W + 2 * pad_w - KW) / stride_w + 1;
    SC2_REQUIRE(OH > 0 && OW > 0, SC2_ERR_INVALID_ARG, "maxpool_nhwc: window larger than the padded map");
    const long long total = (long long)N * OH * OW * (C / 8);
    SC2_REQUIRE((total + 255) / 256 < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "maxpool_nhwc: problem too large");
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(x), static_cast<uint16_t *>(y), H, W, C / 8, OH, OW, KH, KW, stride_h, stride_w, pad_h,
                       pad_w, total);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_avgpool_nhwc(const void *x, float *y_f32, void *y_bf16, int N, int HW, int C, void *stream) {
    SC2_REQUIRE(x && (y_f32 || y_bf16), SC2_ERR_INVALID_ARG, "avgpool_nhwc: null argument");
    SC2_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 8 == 0 && N <= 65535, SC2_ERR_INVALID_ARG,
                "avgpool_nhwc: bad dims N=%d HW=%d C=%d (C %% 8 == 0, N <= 65535)", N, HW, C);
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3((C / 8 + 63) / 64, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(x), y_f32, static_cast<uint16_t *>(y_bf16), HW, C);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_fc_fwd(const void *a, const void *w_frag, const float *bias, float *out, int M, int K, int Npad, void *stream) {
    SC2_REQUIRE(a && w_frag && bias && out, SC2_ERR_INVALID_ARG, "fc: null argument");
    SC2_REQUIRE(M > 0 && K >= 128 && K % 128 == 0 && Npad >= 16 && Npad % 16 == 0, SC2_ERR_INVALID_ARG,
                "fc: bad dims M=%d K=%d Npad=%d (K %% 128 == 0, Npad %% 16 == 0)", M, K, Npad);
    const int gy = (M + FC_RT * 16 - 1) / (FC_RT * 16);
    SC2_REQUIRE(gy <= 65535, SC2_ERR_UNSUPPORTED, "fc: batch too large");
    hipLaunchKernelGGL(fc_kernel, dim3(Npad / 16, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(a), static_cast<const uint16_t *>(w_frag), bias, out, M, K, Npad);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_nhwc_bf16_to_nchw_f32(const void *x, float *y, int N, int C, int H, int W, void *stream) {
    SC2_REQUIRE(x && y, SC2_ERR_INVALID_ARG, "nhwc_to_nchw: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && C % 8 == 0, SC2_ERR_INVALID_ARG,
                "nhwc_to_nchw: bad dims N=%d C=%d H=%d W=%d (C %% 8 != 0)", N, C, H, W);
    const long long total = (long long)N * H * W;
    const int gx = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(gx, C / 8), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(x), y, C, H * W, total);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
