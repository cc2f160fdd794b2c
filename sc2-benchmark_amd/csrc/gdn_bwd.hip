// Element-wise halves of the GDN1 / inverse-GDN1 backward (CompressAI GDN1 under autograd; the reference reaches it
// through loss.backward() for layer.py:478,481,488,491).  With norm = beta + gamma |x| and y = x / norm (inverse: x * norm):
//   pre  : d_norm = -gy * x / norm^2  (inverse: gy * x),   dx_direct = gy / norm  (inverse: gy * norm),
//          d_beta[c] += column sums of d_norm (workgroup partial sums, then one f32 atomic per channel and workgroup)
//   post : dx = dx_direct + sign(x) * t,  t = gamma^T d_norm  (a 1x1 conv on the implicit-GEMM kernel)
// The two channel-mixing products (gamma^T d_norm and d_gamma = d_norm^T |x|) run on conv_igemm / conv_wgrad.
// HBM-bound: 16-byte (8 x bf16) accesses, channel index = element index mod C (NHWC, C % 8 == 0).
#include "sc2_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 r, float *v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[2 * t] = __builtin_bit_cast(float, w[t] << 16);
        v[2 * t + 1] = __builtin_bit_cast(float, w[t] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float *v) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    return o;
}

// grid: (blocks over pixel groups); block 256 threads; thread handles channel chunk cc = tid % (C/8) for a strided set
// of pixels, so its 8 column sums stay in registers.
__global__ __launch_bounds__(256) void gdn_bwd_pre_kernel(const uint16_t *__restrict__ gy, const uint16_t *__restrict__ x,
                                                          const uint16_t *__restrict__ norm, int C, long long M,
                                                          int inverse, uint16_t *__restrict__ d_norm,
                                                          uint16_t *__restrict__ dx_direct, float *__restrict__ d_beta) {
    const int cpr = C >> 3;                       // chunks per pixel
    const int lanes_per_pix = cpr;                // consecutive threads cover one pixel's channels
    const int pix_per_iter = 256 / lanes_per_pix; // pixels a workgroup covers per iteration (cpr divides 256 or not)
    const int cc = threadIdx.x % cpr;
    const int pl = threadIdx.x / cpr;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < pix_per_iter) {
        // two pixels per iteration: six 16-byte loads in flight per thread (the grid is a quarter of what it was, see the launcher)
        const long long stride = (long long)gridDim.x * pix_per_iter;
        for (long long m = (long long)blockIdx.x * pix_per_iter + pl; m < M; m += 2 * stride) {
            const long long o0 = m * C + cc * 8;
            const bool two = m + stride < M;
            const long long o1 = two ? (m + stride) * C + cc * 8 : o0;
            const uint4 rg[2] = {*reinterpret_cast<const uint4 *>(gy + o0), *reinterpret_cast<const uint4 *>(gy + o1)};
            const uint4 rx[2] = {*reinterpret_cast<const uint4 *>(x + o0), *reinterpret_cast<const uint4 *>(x + o1)};
            const uint4 rn[2] = {*reinterpret_cast<const uint4 *>(norm + o0), *reinterpret_cast<const uint4 *>(norm + o1)};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two) break;
                float g[8], xv[8], nv[8], dn[8], dd[8];
                unpack8(rg[h], g);
                unpack8(rx[h], xv);
                unpack8(rn[h], nv);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (inverse) {
                        dn[t] = g[t] * xv[t];
                        dd[t] = g[t] * nv[t];
                    } else {
                        const float r = 1.0f / nv[t];
                        dd[t] = g[t] * r;
                        dn[t] = -dd[t] * xv[t] * r;
                    }
                    sum[t] += dn[t];
                }
                const long long o = h ? o1 : o0;
                *reinterpret_cast<uint4 *>(d_norm + o) = pack8(dn);
                *reinterpret_cast<uint4 *>(dx_direct + o) = pack8(dd);
            }
        }
    }
    // combine the threads that share a channel chunk, then one atomic per channel per workgroup
    __shared__ float red[256][9];
#pragma unroll
    for (int t = 0; t < 8; ++t) red[threadIdx.x][t] = sum[t];
    __syncthreads();
    if (threadIdx.x < cpr) {
        float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = threadIdx.x; q < pix_per_iter * cpr; q += cpr)
#pragma unroll
            for (int t = 0; t < 8; ++t) tot[t] += red[q][t];
#pragma unroll
        for (int t = 0; t < 8; ++t) atomicAdd(d_beta + threadIdx.x * 8 + t, tot[t]);
    }
}

__global__ __launch_bounds__(256) void gdn_bwd_post_kernel(const uint16_t *__restrict__ dx_direct,
                                                           const uint16_t *__restrict__ x, const uint16_t *__restrict__ t,
                                                           long long n_chunks, uint16_t *__restrict__ dx) {
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n_chunks; q += (long long)gridDim.x * 256) {
        float d[8], xv[8], tv[8], o[8];
        unpack8(*reinterpret_cast<const uint4 *>(dx_direct + q * 8), d);
        unpack8(*reinterpret_cast<const uint4 *>(x + q * 8), xv);
        unpack8(*reinterpret_cast<const uint4 *>(t + q * 8), tv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = xv[e] > 0.f ? 1.0f : (xv[e] < 0.f ? -1.0f : 0.f);   // d|x|/dx as torch.abs
            o[e] = d[e] + s * tv[e];
        }
        *reinterpret_cast<uint4 *>(dx + q * 8) = pack8(o);
    }
}

// column sums of a bf16 [M, C] tensor (d_beta of the fused GDN1 backward: sum over pixels of d_norm).  A thread keeps its 8-channel
// chunk for a strided set of pixels; 256 workgroups of 1 024 threads with four 16-byte loads in flight each (16 MB over the chip:
// with 256 threads and two loads the pass ran at 2.8 TB/s, bound by what it kept in flight), partial sums meet in LDS, one f32
// atomic per channel and workgroup.
constexpr int CS_THREADS = 1024, CS_UNROLL = 4;
__global__ __launch_bounds__(CS_THREADS) void colsum_bf16_kernel(const uint16_t *__restrict__ x, long long M, int C, float *__restrict__ out) {
    const int cpr = C >> 3;
    const int ppi = CS_THREADS / cpr;                // pixels a workgroup covers per iteration
    const int cc = threadIdx.x % cpr, pl = threadIdx.x / cpr;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < ppi) {
        const long long stride = (long long)gridDim.x * ppi;
        for (long long m = (long long)blockIdx.x * ppi + pl; m < M; m += CS_UNROLL * stride) {
            uint4 r[CS_UNROLL];
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) {
                const long long mu = m + u * stride;
                r[u] = *reinterpret_cast<const uint4 *>(x + (mu < M ? mu : m) * C + cc * 8);
            }
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) {
                float a[8];
                unpack8(r[u], a);
                const bool live = m + u * stride < M;
#pragma unroll
                for (int t = 0; t < 8; ++t) sum[t] += live ? a[t] : 0.f;
            }
        }
    }
    __shared__ float red[CS_THREADS][9];
#pragma unroll
    for (int t = 0; t < 8; ++t) red[threadIdx.x][t] = sum[t];
    __syncthreads();
    if (threadIdx.x < cpr) {
        float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = threadIdx.x; q < ppi * cpr; q += cpr)
#pragma unroll
            for (int t = 0; t < 8; ++t) tot[t] += red[q][t];
#pragma unroll
        for (int t = 0; t < 8; ++t) atomicAdd(out + threadIdx.x * 8 + t, tot[t]);
    }
}

}  // namespace

extern "C" int sc2_colsum_bf16(const void *x, long long M, int C, float *out, void *stream) {
    SC2_REQUIRE(x && out, SC2_ERR_INVALID_ARG, "colsum_bf16: null argument");
    SC2_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256, SC2_ERR_INVALID_ARG, "colsum_bf16: bad dims M=%lld C=%d", M, C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s);
    SC2_REQUIRE(e == hipSuccess, SC2_ERR_LAUNCH, "colsum_bf16: memset failed: %s", hipGetErrorString(e));
    const int ppi = CS_THREADS / (C / 8);
    long long blocks = (M + ppi - 1) / ppi;
    if (blocks > 256) blocks = 256;      // (one atomic per channel and workgroup on the same C addresses: keep the workgroups few)
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)blocks), dim3(CS_THREADS), 0, s, static_cast<const uint16_t *>(x), M, C, out);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_gdn_bwd_pre(const void *gy, const void *x, const void *norm, long long M, int C, int inverse,
                               void *d_norm, void *dx_direct, float *d_beta, void *stream) {
    SC2_REQUIRE(gy && x && norm && d_norm && dx_direct && d_beta, SC2_ERR_INVALID_ARG, "gdn_bwd_pre: null argument");
    SC2_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256, SC2_ERR_INVALID_ARG, "gdn_bwd_pre: bad dims M=%lld C=%d", M, C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(d_beta, 0, (size_t)C * sizeof(float), s);
    SC2_REQUIRE(e == hipSuccess, SC2_ERR_LAUNCH, "gdn_bwd_pre: memset failed: %s", hipGetErrorString(e));
    const int pix_per_iter = 256 / (C / 8);
    long long blocks = (M + pix_per_iter - 1) / pix_per_iter;
    // every workgroup ends with one f32 atomic per channel on the SAME C addresses: they serialise in L2 (~0.15 us each), so the
    // launch cannot be shorter than (workgroups x that) -- with 4 096 workgroups the 48-channel layer (0.4 GB) took 0.61 ms, as
    // long as layers ten times its size (kernel trace).  1 024 workgroups, two pixels per thread and iteration.
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gdn_bwd_pre_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(gy),
                       static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(norm), C, M, inverse,
                       static_cast<uint16_t *>(d_norm), static_cast<uint16_t *>(dx_direct), d_beta);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_gdn_bwd_post(const void *dx_direct, const void *x, const void *t, long long n_elements, void *dx,
                                void *stream) {
    SC2_REQUIRE(dx_direct && x && t && dx, SC2_ERR_INVALID_ARG, "gdn_bwd_post: null argument");
    SC2_REQUIRE(n_elements > 0 && n_elements % 8 == 0, SC2_ERR_INVALID_ARG, "gdn_bwd_post: element count %% 8 != 0");
    const long long chunks = n_elements / 8;
    long long blocks = (chunks + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gdn_bwd_post_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(dx_direct), static_cast<const uint16_t *>(x),
                       static_cast<const uint16_t *>(t), chunks, static_cast<uint16_t *>(dx));
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
