// GaussianConditional element-wise kernels for gfx950 (hyperprior bottlenecks, sc2bench/models/layer.py:553-817):
// CompressAI 1.2.x `GaussianConditional.forward` / `_likelihood` / `quantize` / `dequantize` / `build_indexes`
// (reached from layer.py:646-647,665,679,691-693,776,785,794,811-813).
//
// All tensors are f32 in NCHW order, `chw` elements per image.  `scales` and `means` may be channel slices of a wider
// tensor (MSHP: gaussian_params.chunk(2, 1)), so they carry their own per-image stride.  HBM-bound: one pass,
// 16-byte accesses where the slice alignment allows, no intermediate tensors (upstream: ~12 element-wise launches).
#include <math.h>

#include "sc2_common.h"

namespace {

struct GcArgs {
    const float *__restrict__ y;
    const float *__restrict__ scales;
    const float *__restrict__ means;   // nullable
    const float *__restrict__ noise;   // nullable (mode NOISE)
    long long n_img, chw, s_stride, m_stride;
    float scale_bound, lik_bound;
    int mode;
};

__device__ __forceinline__ float std_cumulative(float v) {   // 0.5 * erfc(-(2^-0.5) * v), upstream's op order
    const float half = 0.5f;
    const float c = -0.70710678118654752440f;
    return half * erfcf(c * v);
}

__global__ __launch_bounds__(256) void gc_forward_kernel(const GcArgs a, float *__restrict__ y_hat,
                                                         float *__restrict__ lik) {
    const long long total = a.n_img * a.chw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / a.chw, r = i - n * a.chw;
        const float mean = a.means ? a.means[n * a.m_stride + r] : 0.f;
        float out;
        if (a.mode == SC2_EB_NOISE) {
            out = a.y[i] + a.noise[i];               // quantize(x, "noise", means): the means are NOT used
        } else {
            out = rintf(a.y[i] - mean) + mean;       // round-half-even, as torch.round
        }
        if (y_hat) y_hat[i] = out;
        if (lik) {
            const float s = fmaxf(a.scales[n * a.s_stride + r], a.scale_bound);   // LowerBound(scale_bound)
            const float v = fabsf(a.means ? out - mean : out);
            const float upper = std_cumulative((0.5f - v) / s);
            const float lower = std_cumulative((-0.5f - v) / s);
            float p = upper - lower;
            if (a.lik_bound > 0.f) p = fmaxf(p, a.lik_bound);
            lik[i] = p;
        }
    }
}

// symbols = int(round(y - means)); indexes = (n_table - 1) - #{ t < n_table - 1 : max(scales, bound) <= table[t] }
__global__ __launch_bounds__(256) void gc_symbols_indexes_kernel(const GcArgs a, const float *__restrict__ table,
                                                                 int n_table, int32_t *__restrict__ symbols,
                                                                 int32_t *__restrict__ indexes) {
    __shared__ float tab[256];
    for (int t = threadIdx.x; t < n_table; t += 256) tab[t] = table[t];
    __syncthreads();
    const long long total = a.n_img * a.chw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / a.chw, r = i - n * a.chw;
        if (symbols) {
            const float mean = a.means ? a.means[n * a.m_stride + r] : 0.f;
            symbols[i] = (int32_t)rintf(a.y[i] - mean);
        }
        if (indexes) {
            const float s = fmaxf(a.scales[n * a.s_stride + r], a.scale_bound);
            int idx = n_table - 1;
            for (int t = 0; t + 1 < n_table; ++t) idx -= (s <= tab[t]) ? 1 : 0;
            indexes[i] = idx;
        }
    }
}

__global__ __launch_bounds__(256) void gc_dequantize_kernel(const int32_t *__restrict__ symbols,
                                                            const float *__restrict__ means, long long n_img,
                                                            long long chw, long long m_stride, int C, int HW,
                                                            float *__restrict__ y_hat, uint16_t *__restrict__ y_hat_nhwc) {
    const long long total = n_img * chw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / chw, r = i - n * chw;
        const float v = (float)symbols[i] + (means ? means[n * m_stride + r] : 0.f);
        if (y_hat) y_hat[i] = v;
        if (y_hat_nhwc) {
            const int c = (int)(r / HW), pix = (int)(r - (long long)c * HW);
            y_hat_nhwc[(n * HW + pix) * C + c] = f32_to_bf16_bits(v);
        }
    }
}

// Backward of gc_forward_kernel in NOISE mode (the training mode: y_hat = y + noise, so d y_hat / d y = 1 and the means
// enter only through v = |y_hat - mean|).  With u = (.5 - v)/s, l = (-.5 - v)/s, p = Phi(u) - Phi(l):
//   dp/dv = (phi(l) - phi(u)) / s        dp/ds = (l phi(l) - u phi(u)) / s        phi = standard normal density
// LowerBound gradients as upstream: through the likelihood bound where p >= bound or the gradient is negative, through
// the scale bound where scales >= bound or the gradient is negative.
__global__ __launch_bounds__(256) void gc_backward_kernel(const GcArgs a, const float *__restrict__ g_yhat,
                                                          const float *__restrict__ g_lik, float *__restrict__ g_y,
                                                          float *__restrict__ g_scales, float *__restrict__ g_means) {
    const long long total = a.n_img * a.chw;
    const float inv_sqrt_2pi = 0.39894228040143267794f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / a.chw, r = i - n * a.chw;
        const float mean = a.means ? a.means[n * a.m_stride + r] : 0.f;
        const float out = a.y[i] + (a.noise ? a.noise[i] : 0.f);
        const float s_raw = a.scales[n * a.s_stride + r];
        const float s = fmaxf(s_raw, a.scale_bound);
        const float dlt = out - mean;
        const float v = fabsf(dlt);
        const float u = (0.5f - v) / s, l = (-0.5f - v) / s;
        const float p = std_cumulative(u) - std_cumulative(l);
        float gp = g_lik ? g_lik[i] : 0.f;
        if (a.lik_bound > 0.f && !(p >= a.lik_bound || gp < 0.f)) gp = 0.f;
        const float phi_u = inv_sqrt_2pi * expf(-0.5f * u * u), phi_l = inv_sqrt_2pi * expf(-0.5f * l * l);
        const float dp_dv = (phi_l - phi_u) / s;
        const float dp_ds = (l * phi_l - u * phi_u) / s;
        const float sgn = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
        const float g_from_lik = gp * dp_dv * sgn;
        if (g_y) g_y[i] = (g_yhat ? g_yhat[i] : 0.f) + g_from_lik;
        if (g_means) g_means[i] = -g_from_lik;
        if (g_scales) {
            float gs = gp * dp_ds;
            if (!(s_raw >= a.scale_bound || gs < 0.f)) gs = 0.f;
            g_scales[i] = gs;
        }
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int sc2_gc_forward(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                              int64_t means_img_stride, const float *noise, int64_t n_img, int64_t chw, int mode,
                              float scale_bound, float lik_bound, float *y_hat, float *lik, void *stream) {
    SC2_REQUIRE(y && (y_hat || lik), SC2_ERR_INVALID_ARG, "gc_forward: null argument");
    SC2_REQUIRE(!lik || scales, SC2_ERR_INVALID_ARG, "gc_forward: likelihoods need scales");
    SC2_REQUIRE(n_img > 0 && chw > 0, SC2_ERR_INVALID_ARG, "gc_forward: bad dims");
    SC2_REQUIRE(mode == SC2_EB_NOISE || mode == SC2_EB_DEQUANTIZE, SC2_ERR_INVALID_ARG,
                "Invalid quantization mode: \"%d\"", mode);
    if (mode == SC2_EB_NOISE) SC2_REQUIRE(noise, SC2_ERR_INVALID_ARG, "gc_forward: noise mode needs a noise tensor");
    SC2_REQUIRE(scale_bound > 0.f, SC2_ERR_INVALID_ARG, "gc_forward: scale_bound must be positive");
    GcArgs a;
    a.y = y; a.scales = scales; a.means = means; a.noise = noise;
    a.n_img = n_img; a.chw = chw; a.s_stride = scales_img_stride; a.m_stride = means_img_stride;
    a.scale_bound = scale_bound; a.lik_bound = lik_bound; a.mode = mode;
    hipLaunchKernelGGL(gc_forward_kernel, dim3(grid_for(n_img * chw)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       y_hat, lik);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_gc_symbols_indexes(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                                      int64_t means_img_stride, int64_t n_img, int64_t chw, const float *scale_table,
                                      int n_table, float scale_bound, int32_t *symbols, int32_t *indexes, void *stream) {
    SC2_REQUIRE(symbols || indexes, SC2_ERR_INVALID_ARG, "gc_symbols_indexes: no output");
    SC2_REQUIRE(!symbols || y, SC2_ERR_INVALID_ARG, "gc_symbols_indexes: symbols need y");
    SC2_REQUIRE(!indexes || (scales && scale_table), SC2_ERR_INVALID_ARG, "gc_symbols_indexes: indexes need scales");
    SC2_REQUIRE(n_img > 0 && chw > 0, SC2_ERR_INVALID_ARG, "gc_symbols_indexes: bad dims");
    SC2_REQUIRE(!indexes || (n_table >= 1 && n_table <= 256), SC2_ERR_UNSUPPORTED,
                "gc_symbols_indexes: scale table of %d entries (1..256 supported)", n_table);
    GcArgs a;
    a.y = y; a.scales = scales; a.means = means; a.noise = nullptr;
    a.n_img = n_img; a.chw = chw; a.s_stride = scales_img_stride; a.m_stride = means_img_stride;
    a.scale_bound = scale_bound; a.lik_bound = 0.f; a.mode = SC2_EB_DEQUANTIZE;
    hipLaunchKernelGGL(gc_symbols_indexes_kernel, dim3(grid_for(n_img * chw)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, scale_table, n_table, symbols, indexes);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_gc_dequantize(const int32_t *symbols, const float *means, int64_t means_img_stride, int64_t n_img,
                                 int C, int HW, float *y_hat_f32_nchw, void *y_hat_bf16_nhwc, void *stream) {
    SC2_REQUIRE(symbols && (y_hat_f32_nchw || y_hat_bf16_nhwc), SC2_ERR_INVALID_ARG, "gc_dequantize: null argument");
    SC2_REQUIRE(n_img > 0 && C > 0 && HW > 0, SC2_ERR_INVALID_ARG, "gc_dequantize: bad dims");
    const long long chw = (long long)C * HW;
    hipLaunchKernelGGL(gc_dequantize_kernel, dim3(grid_for(n_img * chw)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       symbols, means, (long long)n_img, chw, (long long)means_img_stride, C, HW, y_hat_f32_nchw,
                       static_cast<uint16_t *>(y_hat_bf16_nhwc));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_gc_backward(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                               int64_t means_img_stride, const float *noise, int64_t n_img, int64_t chw,
                               float scale_bound, float lik_bound, const float *g_yhat, const float *g_lik, float *g_y,
                               float *g_scales, float *g_means, void *stream) {
    SC2_REQUIRE(y && scales && (g_y || g_scales || g_means), SC2_ERR_INVALID_ARG, "gc_backward: null argument");
    SC2_REQUIRE(n_img > 0 && chw > 0 && scale_bound > 0.f, SC2_ERR_INVALID_ARG, "gc_backward: bad arguments");
    SC2_REQUIRE(!g_means || means, SC2_ERR_INVALID_ARG, "gc_backward: a means gradient needs means");
    GcArgs a;
    a.y = y; a.scales = scales; a.means = means; a.noise = noise;
    a.n_img = n_img; a.chw = chw; a.s_stride = scales_img_stride; a.m_stride = means_img_stride;
    a.scale_bound = scale_bound; a.lik_bound = lik_bound; a.mode = SC2_EB_NOISE;
    hipLaunchKernelGGL(gc_backward_kernel, dim3(grid_for(n_img * chw)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       g_yhat, g_lik, g_y, g_scales, g_means);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
