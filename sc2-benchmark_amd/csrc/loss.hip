// Element-wise pieces of the distillation step that torch would run as several passes over 100 - 800 MB tensors:
//   * sum((x - y)^2) of two bf16 tensors (nn.MSELoss(reduction='sum') of the reference's stage-1 recipe:
//     configs/ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_resnet50.yaml:155-200,
//     four feature maps per step, the largest 205 M elements), f32 accumulation, one pass over both operands;
//   * its input gradient 2 (x - y) * scale as bf16, one pass;
//   * the gradient through a fused ReLU: (g [+ a second branch]) * (out > 0), one pass (the mask comes from the saved OUTPUT).
// HBM-bound streaming kernels: 16-byte loads / stores, grid-stride.
#include "sc2_common.h"

namespace {

__device__ __forceinline__ float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

__global__ __launch_bounds__(256) void mse_sum_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ y, long long n8,
                                                      float *__restrict__ partial) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 a = x[i], b = y[i];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d0 = lo(aw[k]) - lo(bw[k]), d1 = hi(aw[k]) - hi(bw[k]);
            acc = fmaf(d0, d0, acc);
            acc = fmaf(d1, d1, acc);
        }
    }
    // fixed-order reduction inside the block: lanes by xor-shuffle, then the four waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void mse_grad_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ y, long long n8,
                                                       const float *__restrict__ scale, uint4 *__restrict__ gx) {
    const float s = 2.0f * scale[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 a = x[i], b = y[i];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(s * (lo(aw[k]) - lo(bw[k])), s * (hi(aw[k]) - hi(bw[k])));
        gx[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// g_in = (g [+ add]) * (out > 0): the gradient through a ReLU fused into a conv epilogue; `add` = a second gradient that
// reaches the same tensor (the two branches of a Bottleneck block -- conv1's data gradient and the identity -- meet at the
// previous block's output), summed in f32 and rounded once
__global__ __launch_bounds__(256) void relu_bwd_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ out,
                                                       const uint4 *__restrict__ add, long long n8, uint4 *__restrict__ gi) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 a = g[i], m = out[i];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, mw[4] = {m.x, m.y, m.z, m.w};
        uint32_t o[4];
        if (add) {
            const uint4 c = add[i];
            const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2(lo(mw[k]) > 0.f ? lo(aw[k]) + lo(cw[k]) : 0.f, hi(mw[k]) > 0.f ? hi(aw[k]) + hi(cw[k]) : 0.f);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = (lo(mw[k]) > 0.f ? aw[k] & 0xFFFFu : 0u) | (hi(mw[k]) > 0.f ? aw[k] & 0xFFFF0000u : 0u);
        }
        gi[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// The same with the MSE term that sits on this tensor folded in (round 5): g_in = ([g] + 2 scale (out - t)) * (out > 0).  `out` is the
// stack output that the feature-matching loss compares with the teacher's t, so the loss gradient needs no operand of its own: the
// three passes mse_grad (x, y -> g_mse), add (g, g_mse -> sum), relu_bwd (sum, out -> g_in) = nine tensor reads / writes become four.
// One rounding instead of three.
template <bool RELU>   // false: no activation behind the tensor (the bottleneck's own output): g_in = [g] + 2 scale (out - t)
__global__ __launch_bounds__(256) void relu_bwd_mse_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ out,
                                                           const uint4 *__restrict__ t, const float *__restrict__ scale, long long n8,
                                                           uint4 *__restrict__ gi) {
    const float s = 2.0f * scale[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 m = out[i], b = t[i];
        const uint4 a = g ? g[i] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, mw[4] = {m.x, m.y, m.z, m.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16x2((!RELU || lo(mw[k]) > 0.f) ? lo(aw[k]) + s * (lo(mw[k]) - lo(bw[k])) : 0.f,
                               (!RELU || hi(mw[k]) > 0.f) ? hi(aw[k]) + s * (hi(mw[k]) - hi(bw[k])) : 0.f);
        gi[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
int grid_for(long long n8) {
    const long long b = (n8 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int sc2_mse_partial_len(long long n) { return n > 0 ? grid_for(n / 8) : 0; }

extern "C" int sc2_mse_sum_bf16(const void *x, const void *y, long long n, float *partial, void *stream) {
    SC2_REQUIRE(x && y && partial, SC2_ERR_INVALID_ARG, "mse_sum: null argument");
    SC2_REQUIRE(n > 0 && n % 8 == 0, SC2_ERR_INVALID_ARG, "mse_sum: element count %lld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(mse_sum_kernel, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(x), static_cast<const uint4 *>(y), n / 8, partial);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_mse_grad_bf16(const void *x, const void *y, long long n, const float *scale, void *gx, void *stream) {
    SC2_REQUIRE(x && y && scale && gx, SC2_ERR_INVALID_ARG, "mse_grad: null argument");
    SC2_REQUIRE(n > 0 && n % 8 == 0, SC2_ERR_INVALID_ARG, "mse_grad: element count %lld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(x), static_cast<const uint4 *>(y), n / 8, scale, static_cast<uint4 *>(gx));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_relu_bwd_bf16(const void *g, const void *out, const void *add, long long n, void *gi, void *stream) {
    SC2_REQUIRE(g && out && gi, SC2_ERR_INVALID_ARG, "relu_bwd: null argument");
    SC2_REQUIRE(n > 0 && n % 8 == 0, SC2_ERR_INVALID_ARG, "relu_bwd: element count %lld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(g), static_cast<const uint4 *>(out), static_cast<const uint4 *>(add), n / 8,
                       static_cast<uint4 *>(gi));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
extern "C" int sc2_relu_bwd_mse_bf16(const void *g, const void *out, const void *t, const float *scale, long long n, int relu, void *gi,
                                     void *stream) {
    SC2_REQUIRE(out && t && scale && gi, SC2_ERR_INVALID_ARG, "relu_bwd_mse: null argument");
    SC2_REQUIRE(n > 0 && n % 8 == 0, SC2_ERR_INVALID_ARG, "relu_bwd_mse: element count %lld must be a positive multiple of 8", n);
    if (relu)
        hipLaunchKernelGGL(relu_bwd_mse_kernel<true>, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint4 *>(g), static_cast<const uint4 *>(out), static_cast<const uint4 *>(t), scale, n / 8,
                           static_cast<uint4 *>(gi));
    else
        hipLaunchKernelGGL(relu_bwd_mse_kernel<false>, dim3(grid_for(n / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint4 *>(g), static_cast<const uint4 *>(out), static_cast<const uint4 *>(t), scale, n / 8,
                           static_cast<uint4 *>(gi));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
