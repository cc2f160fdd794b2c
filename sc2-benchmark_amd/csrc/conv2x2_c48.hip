// The last encoder convolution of the FP bottleneck (gfx950): Conv2d(48 -> 24, k2, s1, p0, bias=False) on the GDN1(48) output,
// sc2bench/models/layer.py:482 (`encoder[4]`), writing what the entropy bottleneck reads: the f32 NCHW latent, or (evaluation
// after update()) the int32 symbols round-half-even(y - median) of EntropyModel.quantize(..., 'symbols') directly
// (layer.py:496-507 -> compress).  y[n, co, oh, ow] = sum_{kh, kw, ci} x[n, oh + kh, ow + kw, ci] w[co, ci, kh, kw].
//
// Why a dedicated kernel.  27.9 MFLOP per image against 0.30 MB in + 0.29 MB out: the layer is HBM-bound (151 MB per 256-image
// batch = ~0.03 ms), but as 6 050 four-wave tiles of the implicit-GEMM kernel (six k-slabs through LDS, an LDS-staged NCHW
// epilogue per 128 pixels) it took 0.052 ms alone and 0.11 - 0.18 ms inside the pipeline, where launches made of many short
// workgroups suffer most from the kernels running beside them.  Here
//   * in NHWC the two taps (kh, 0), (kh, 1) of an output pixel are 96 CONTIGUOUS channels (two neighbouring pixels): K = 192 is two
//     runs of three 32-deep k-steps whose MFMA operand fragments are plain 16-byte global loads -- no LDS, no barrier;
//   * the weights (24 x 192, zero rows to 32) stay in registers for the life of the wave (12 fragments);
//   * a wave walks 16-pixel tiles of the flattened (n, oh, ow) axis with the next tile's six loads in flight; the grid is sized by
//     the chip, not by the problem.
// Same k order (kh, kw, ci), same MFMA instruction and operand roles as the tile kernel: results are bit-identical to
// sc2_conv2d_fwd's (tests/test_gpu_kernels.py::test_conv2x2_c48).
#include "sc2_common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t, uint32_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
#endif

struct C48Args {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, W, 48]
    const uint16_t *__restrict__ w;      // bf16 fragment blocks [2][6][64][8] of the [32, 192] matrix W[co][kh*96 + kw*48 + ci]
    const float *__restrict__ med;       // f32 [Cout] medians (symbol mode) or null
    void *__restrict__ y;                // f32 or int32 NCHW [N, Cout, OH, OW]
    int N, H, W, OH, OW, Cout, sym;
    int n_tiles;                         // ceil(N * OH * OW / 16)
    unsigned x_bytes;
};

__global__ __launch_bounds__(256) void conv2x2_c48_kernel(const C48Args p) {
    constexpr uint32_t OOB = 0x80000000u;
    const int lane = threadIdx.x & 63;
    const int frow = lane & 15, fq = lane >> 4;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const buf_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
    const int OHW = p.OH * p.OW;
    const long long M = (long long)p.N * OHW;

    // resident weight fragments: [channel tile t][k-step j] -> lane's 8 k-values of channel row 16 t + frow
    bf16x8_t wf[2][6];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wf[t][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(p.w + ((t * 6 + j) * 64 + lane) * 8));
    // this lane's output channels: 16 t + 4 fq + r
    float med[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * fq + r;
            med[t][r] = (p.sym && c < p.Cout) ? p.med[c] : 0.f;
        }

    // byte offset of x[n, oh, ow, 8 fq] for this lane's pixel of tile `tile` (OOB past the end: zeros)
    auto pixel_of = [&](int tile, int &img, int &pix) -> uint32_t {
        const long long m = (long long)tile * 16 + frow;
        if (m >= M) { img = 0; pix = -1; return OOB; }
        img = (int)(m / OHW);
        pix = (int)(m - (long long)img * OHW);
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        return (uint32_t)((((img * p.H + oh) * p.W + ow) * 48 + fq * 8) * 2);
    };
    const uint32_t row_bytes = (uint32_t)p.W * 96u;   // one input row down (kh = 1)
    auto load_tile = [&](uint32_t vo, uint4 (&a)[6]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[kh * 3 + j] = buf_load16(rs_x, vo, (uint32_t)kh * row_bytes + (uint32_t)j * 64u);
    };

    int tile = wave_id;
    if (tile >= p.n_tiles) return;
    int img, pix;
    uint4 a_nxt[6];
    load_tile(pixel_of(tile, img, pix), a_nxt);
    while (true) {
        uint4 a[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = a_nxt[j];
        const int img_c = img, pix_c = pix;
        const int nxt = tile + n_waves;
        if (nxt < p.n_tiles) load_tile(pixel_of(nxt, img, pix), a_nxt);

        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, a[j]);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af, acc[1], 0, 0, 0);
        }
        if (pix_c >= 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * fq + r;
                    if (c < p.Cout) {
                        const long long o = ((long long)img_c * p.Cout + c) * OHW + pix_c;
                        if (p.sym) reinterpret_cast<int32_t *>(p.y)[o] = (int32_t)rintf(acc[t][r] - med[t][r]);
                        else reinterpret_cast<float *>(p.y)[o] = acc[t][r];
                    }
                }
        }
        if (nxt >= p.n_tiles) break;
        tile = nxt;
    }
}

}  // namespace

extern "C" int sc2_conv2x2_c48_supported(int H, int W, int Cin, int Cout) {
    return Cin == 48 && Cout >= 1 && Cout <= 32 && H >= 2 && W >= 2 ? 1 : 0;
}

extern "C" int sc2_conv2x2_c48_fwd(const void *x, const void *w_frag, const float *medians, void *y, int N, int H, int W, int Cin,
                                   int Cout, int symbols, void *stream) {
    SC2_REQUIRE(x && w_frag && y, SC2_ERR_INVALID_ARG, "conv2x2_c48: null argument");
    SC2_REQUIRE(N > 0, SC2_ERR_INVALID_ARG, "conv2x2_c48: non-positive batch");
    SC2_REQUIRE(sc2_conv2x2_c48_supported(H, W, Cin, Cout), SC2_ERR_UNSUPPORTED,
                "conv2x2_c48: needs Cin == 48, Cout <= 32 and a map of at least 2 x 2 (got %d x %d, %d -> %d)", H, W, Cin, Cout);
    SC2_REQUIRE(!symbols || medians, SC2_ERR_INVALID_ARG, "conv2x2_c48: symbol output needs the medians");
    const long long x_bytes = (long long)N * H * W * 48 * 2;
    const long long M = (long long)N * (H - 1) * (W - 1);
    SC2_REQUIRE(x_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv2x2_c48: operand of %lld bytes exceeds 2 GB", x_bytes);
    C48Args a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.med = medians;
    a.y = y;
    a.N = N; a.H = H; a.W = W; a.OH = H - 1; a.OW = W - 1; a.Cout = Cout; a.sym = symbols ? 1 : 0;
    a.n_tiles = (int)((M + 15) / 16);
    a.x_bytes = (unsigned)x_bytes;
    // four workgroups of four waves per CU, never more waves than tiles
    const int n_cu = sc2_device_cus();
    int grid = n_cu * 4;
    if ((long long)grid * 4 > a.n_tiles) grid = (a.n_tiles + 3) / 4;
    hipLaunchKernelGGL(conv2x2_c48_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
