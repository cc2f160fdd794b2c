// Weight gradient of the implicit-GEMM convolution for gfx950:
//
//   dW[co][k] = sum_m gY[m][co] * A[m][k]      m = (img, oh, ow),  k = (kh, kw, ci),  A = im2col(x)
//
// i.e. the training-time counterpart of nn.Conv2d's backward w.r.t. its weight (the reference reaches it through
// loss.backward() in script/task/image_classification.py:79 for the convs of sc2bench/models/layer.py:475-493, and
// for GDN1's 1x1 gamma).  It is a GEMM whose REDUCTION index is the pixel index m, while both operands are stored
// pixel-major (NHWC rows).  Instead of transposing on the way into LDS, the slabs keep their natural [m][channel]
// layout (so they are filled by plain direct-to-LDS loads) and the MFMA fragments are read with the transposing LDS
// read ds_read_b64_tr_b16 (4 rows x 16 columns -> column-major), two reads per 16x16x32 operand fragment.
//
// Tile: 128 output channels x 128 k-columns per workgroup (4 waves, 64 x 64 each), reduction in slabs of 32 pixels
// through a 3-deep LDS ring (same counted-vmcnt / one-barrier-per-slab pipeline as conv_igemm.hip).  The pixel range
// is split over workgroups; partial sums are combined with f32 atomic adds into the (small, pre-zeroed) dW.
// Bank conflicts: a fragment read touches 8 different pixel rows at one 32-byte column offset; the 16-byte chunk
// index is XORed with f(row) = 2*((row & 3) + 4*((row >> 3) & 1)) -- applied to the SOURCE address of the
// direct-to-LDS load and again to the read address -- so the 8 rows land on 8 different 32-byte slots.
#include <stdlib.h>

#include <type_traits>

#include "sc2_common.h"

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ uint4 g_wzero16;

struct WgradArgs {
    const uint16_t *__restrict__ x;
    const uint16_t *__restrict__ gy;
    float *__restrict__ dw;
    int N, H, W, Cin, Cout;
    int KH, KW, SH, SW, PH, PW;
    int OH, OW, OHW, M, K;
    int rows_per_block;   // pixels reduced by one workgroup (multiple of 32)
    int x_abs;            // use |x| as the im2col operand (d gamma of GDN1)
    int n_ktiles, n_ctiles;
};

constexpr int WG_TILE = 128;              // channels / k-columns per tile
constexpr int WG_ROWB = WG_TILE * 2;      // bytes per pixel row of a slab image
constexpr int WG_SLAB = 32;               // pixels per slab
constexpr int WG_IMG = WG_SLAB * WG_ROWB; // 8 KB
constexpr int WG_STAGES = 3;

__device__ __forceinline__ int wg_swz(int row) { return 2 * ((row & 3) + 4 * ((row >> 3) & 1)); }

template <int OFF>
__device__ __forceinline__ uint2 lds_read_tr(uint32_t addr) {   // (stage base as an immediate: no address arithmetic per read)
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <bool ABS>   // |x| as the im2col operand (d gamma of GDN1): a compile-time property (16 mask operations per slab otherwise)
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // Workgroup b runs on XCD b & 7 (round-robin dispatch).  The `tiles` workgroups that reduce the SAME pixel range -- each reads the
    // range's dY slab for its 128 output channels and the im2col slab for its 128 k-columns -- are consecutive workgroups of ONE XCD:
    // the range's rows of x and dY then come from HBM once, into that XCD's L2, instead of once per XCD (round 5: with the ranges dealt
    // out in launch order a range's 16 - 32 tiles were spread over all eight L2s).  The launcher makes the number of ranges a multiple
    // of 8; range c belongs to XCD c & 7.
    const int tiles = p.n_ktiles * p.n_ctiles;
    const int xcd = (int)(blockIdx.x & 7), local = (int)(blockIdx.x >> 3);
    const int chunk = (local / tiles) * 8 + xcd;
    const int t = local % tiles;
    if (chunk * p.rows_per_block >= p.M) return;   // (a range past the end, from rounding the range count up)
    const int ctile = t / p.n_ktiles, ktile = t - ctile * p.n_ktiles;
    const int co0 = ctile * WG_TILE, k0 = ktile * WG_TILE;
    const int m_begin = chunk * p.rows_per_block;
    const int m_end = min(p.M, m_begin + p.rows_per_block);
    const int n_slabs = (m_end - m_begin + WG_SLAB - 1) / WG_SLAB;

    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_wzero16);
    const long long zoff_x = zero - p.x, zoff_g = zero - p.gy;

    // direct-to-LDS assignment: wave-instruction q = j*4 + wave covers slab rows [4q, 4q+4); lane l -> row 4q + (l >> 4),
    // stored chunk l & 15, which holds logical chunk (l & 15) ^ swz(row)
    int row_j[2], gco_j[2], kh_j[2], kw_j[2], ci_j[2];
    bool gok_j[2], kok_j[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 4 + wave) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ wg_swz(row);
        row_j[j] = row;
        gco_j[j] = co0 + 8 * c;
        gok_j[j] = gco_j[j] < p.Cout;
        const int k = k0 + 8 * c;
        kok_j[j] = k < p.K;
        const int kk = kok_j[j] ? k : 0;
        const int tap = kk / p.Cin;
        ci_j[j] = kk - tap * p.Cin;
        kh_j[j] = tap / p.KW;
        kw_j[j] = tap - kh_j[j] * p.KW;
    }

    // Pixel coordinates of this lane's two slab rows, carried from slab to slab (round 4).  Slabs are issued strictly in order,
    // each 32 pixels further; recomputing (image, oh, ow) from the pixel index cost two runtime divisions per piece and slab --
    // 183 vector instructions per slab against its 16 MFMAs (listing): the kernel was bound by its address arithmetic.  Now a
    // slab step adds the decomposition of 32 pixels (d_img, d_oh, d_ow) with one conditional subtract per coordinate.
    const int d_img = WG_SLAB / p.OHW, d_rem = WG_SLAB - d_img * p.OHW;
    const int d_oh = d_rem / p.OW, d_ow = d_rem - d_oh * p.OW;
    // ... and so are the input coordinates (ih, iw) of the lane's tap and the ELEMENT OFFSETS of both operands (second pass: the
    // loop still multiplied -- oh * SH, ow * SW, ((img * H + ih) * W + iw) * Cin: ten quarter-rate instructions per slab beside
    // sixty others against 16 MFMAs).  A slab step moves the im2col offset by cA, plus cB when the output column wraps, plus cC
    // when the output row does (32-bit modular arithmetic; the offset is used only where the tap lies inside the image).
    const uint32_t g_step = (uint32_t)(WG_SLAB * p.Cout);
    const int iw_step = d_ow * p.SW, iw_wrap = p.OW * p.SW, ih_step = d_oh * p.SH, ih_wrap = p.OH * p.SH;
    const uint32_t cA = (uint32_t)(p.Cin * (p.W * d_oh * p.SH + d_ow * p.SW) + p.Cin * p.H * p.W * d_img);
    const uint32_t cB = (uint32_t)(p.Cin * (p.W * p.SH - p.OW * p.SW));
    const uint32_t cC = (uint32_t)(p.Cin * (p.H * p.W - p.W * p.OH * p.SH));
    int m_j[2], oh_j[2], ow_j[2], ih_j[2], iw_j[2];
    uint32_t goff_j[2], aoff_j[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        m_j[j] = m_begin + row_j[j];
        const int mm = m_j[j] < p.M ? m_j[j] : 0;      // (rows past the end: coordinates of pixel 0, never used)
        const int img = mm / p.OHW;
        const int rem = mm - img * p.OHW;
        oh_j[j] = rem / p.OW;
        ow_j[j] = rem - oh_j[j] * p.OW;
        ih_j[j] = oh_j[j] * p.SH - p.PH + kh_j[j];
        iw_j[j] = ow_j[j] * p.SW - p.PW + kw_j[j];
        goff_j[j] = (uint32_t)(m_j[j] * p.Cout + gco_j[j]);
        aoff_j[j] = (uint32_t)(((img * p.H + ih_j[j]) * p.W + iw_j[j]) * p.Cin + ci_j[j]);
    }
    auto issue_slab = [&](int buf) {   // the NEXT slab in order
        unsigned char *Gi = smem + buf * (2 * WG_IMG);
        unsigned char *Ai = Gi + WG_IMG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool mok = m_j[j] < m_end;
            // dY operand
            const bool g_ok = mok & gok_j[j];
            const long long goff = g_ok ? (long long)goff_j[j] : zoff_g;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.gy + goff), (lds_ptr_t)(Gi + (j * 4 + wave) * 1024), 16, 0, 0);
            // im2col operand
            const bool a_ok = mok & kok_j[j] & ((unsigned)ih_j[j] < (unsigned)p.H) & ((unsigned)iw_j[j] < (unsigned)p.W);
            const long long aoff = a_ok ? (long long)aoff_j[j] : zoff_x;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.x + aoff), (lds_ptr_t)(Ai + (j * 4 + wave) * 1024), 16, 0, 0);
            // 32 pixels on
            m_j[j] += WG_SLAB;
            goff_j[j] += g_step;
            ow_j[j] += d_ow;
            const bool c1 = ow_j[j] >= p.OW;
            ow_j[j] -= c1 ? p.OW : 0;
            iw_j[j] += iw_step - (c1 ? iw_wrap : 0);
            oh_j[j] += d_oh + (c1 ? 1 : 0);
            const bool c2 = oh_j[j] >= p.OH;
            oh_j[j] -= c2 ? p.OH : 0;
            ih_j[j] += ih_step + (c1 ? p.SH : 0) - (c2 ? ih_wrap : 0);
            aoff_j[j] += cA + (c1 ? cB : 0u) + (c2 ? cC : 0u);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposing fragment reads: lane i16 of a 16-lane group addresses slab row 8*fq + (i16 >> 2) (+4 for the second
    // half of the 8-deep k group) at columns 4*(i16 & 3) .. +3 of the 16-column tile and receives column i16
    const int i16 = lane & 15, fq = lane >> 4;
    const int rrow = 8 * fq + (i16 >> 2);
    uint32_t g_rd[4][2], a_rd[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rrow + 4 * h;
            const int gc = wm * 64 + tt * 16 + 4 * (i16 & 3);   // column inside the 128-wide image
            const int ac = wn * 64 + tt * 16 + 4 * (i16 & 3);
            g_rd[tt][h] = lds_base + (uint32_t)(r * WG_ROWB + (((gc >> 3) ^ wg_swz(r)) << 4) + ((gc >> 2) & 1) * 8);
            a_rd[tt][h] = lds_base + (uint32_t)(r * WG_ROWB + (((ac >> 3) ^ wg_swz(r)) << 4) + ((ac >> 2) & 1) * 8);
        }

    constexpr uint32_t xmask = 0x7FFF7FFFu;
    constexpr int S = WG_STAGES, L = 4;
    static_assert(S == 3, "the slab loop below is unrolled by the ring depth");
#pragma unroll
    for (int st = 0; st < S - 1; ++st) issue_slab(st);

    // one slab out of ring stage ST (compile-time: the fragment reads address the stage through an immediate offset)
    auto slab = [&](auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        constexpr int GOFF = ST * (2 * WG_IMG), AOFF = GOFF + WG_IMG;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        uint2 gv[4][2], av[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gv[tt][h] = lds_read_tr<GOFF>(g_rd[tt][h]);
                av[tt][h] = lds_read_tr<AOFF>(a_rd[tt][h]);
            }
        issue_slab((ST + S - 1) % S);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t gf[4], af[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            gf[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(gv[tt][0].x, gv[tt][0].y, gv[tt][1].x, gv[tt][1].y));
            if (ABS) af[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(av[tt][0].x & xmask, av[tt][0].y & xmask, av[tt][1].x & xmask,
                                                                      av[tt][1].y & xmask));
            else af[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(av[tt][0].x, av[tt][0].y, av[tt][1].x, av[tt][1].y));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], af[j], acc[i][j], 0, 0, 0);
    };
    int sl = 0;
    for (; sl + S <= n_slabs; sl += S) {
        slab(std::integral_constant<int, 0>{});
        slab(std::integral_constant<int, 1>{});
        slab(std::integral_constant<int, 2>{});
    }
    if (sl < n_slabs) slab(std::integral_constant<int, 0>{});
    if (sl + 1 < n_slabs) slab(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // combine the pixel-range partial sums: f32 atomic adds (dW is small; arrival order varies run to run)
    const int frow = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wm * 64 + i * 16 + fq * 4 + e;
                const int k = k0 + wn * 64 + j * 16 + frow;
                if (co < p.Cout && k < p.K) atomicAdd(p.dw + (long long)co * p.K + k, acc[i][j][e]);
            }
}

}  // namespace

extern "C" int sc2_conv2d_wgrad(const sc2_conv_desc *d, const void *x, const void *gy, float *dw, void *stream) {
    SC2_REQUIRE(d && x && gy && dw, SC2_ERR_INVALID_ARG, "conv2d_wgrad: null argument");
    SC2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Cin % 8 == 0 && d->Cout % 8 == 0,
                SC2_ERR_INVALID_ARG, "conv2d_wgrad: bad dims (Cin %d, Cout %d must be multiples of 8)", d->Cin, d->Cout);
    SC2_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                SC2_ERR_INVALID_ARG, "conv2d_wgrad: bad filter geometry");
    const int OH = (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1;
    const int OW = (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1;
    SC2_REQUIRE(OH == d->OH && OW == d->OW && OH > 0 && OW > 0, SC2_ERR_INVALID_ARG,
                "conv2d_wgrad: output size %dx%d does not match geometry (%dx%d)", d->OH, d->OW, OH, OW);
    const long long M = (long long)d->N * OH * OW;
    SC2_REQUIRE(M < 0x7FFFFFFFLL - 4096 && (long long)d->N * d->H < 0x7FFFFFFFLL && M * d->Cout < 0xFFFFFFFFLL &&
                    (long long)d->N * d->H * d->W * d->Cin < 0xFFFFFFFFLL,
                SC2_ERR_UNSUPPORTED, "conv2d_wgrad: problem too large (32-bit element offsets)");
    WgradArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.gy = static_cast<const uint16_t *>(gy);
    a.dw = dw;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.SH = d->stride_h; a.SW = d->stride_w; a.PH = d->pad_h; a.PW = d->pad_w;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW; a.M = (int)M; a.K = d->KH * d->KW * d->Cin;
    a.x_abs = d->a_op == SC2_AOP_ABS;
    a.n_ktiles = (a.K + WG_TILE - 1) / WG_TILE;
    a.n_ctiles = (a.Cout + WG_TILE - 1) / WG_TILE;
    // enough pixel chunks to fill the chip several times over, at least 8 slabs each
    const long long tiles = (long long)a.n_ktiles * a.n_ctiles;
    // workgroups per launch = output tiles x pixel ranges.  Every workgroup ends with 128 x 128 f32 atomic adds, so the pixel split
    // is paid in atomics: 4 096 workgroups (16 per CU) -> 1 024 (two rounds of the 512 resident ones) took the small layers'
    // launches from 0.095 / 0.31 / 0.28 ms to 0.054 / 0.24 / 0.24 (the 512 x 512 gamma gradient stays at 0.89: it is bound by
    // the L2 -> LDS fill of its 128 x 128 tiles, 16 KB per 64 MFMAs), the training step + 2 %.  SC2_WGRAD_WGS overrides (A/B).
    // (round 5, ranges XCD-local -- tools/wgrad_times.py: four output tiles or fewer want 512 workgroups (igdn3's gamma 0.177 -> 0.152
    //  ms), a single 128-channel tile row with a long K -- enc.conv2, 19 k-tiles -- wants 4 096 (0.63 -> 0.53); the rest is flat
    //  between 1 024 and 2 048)
    const int wg_auto = tiles <= 4 ? 512 : (a.n_ctiles == 1 ? 4096 : 1024);
    const int wg_target = sc2_pol().wgrad_wgs > 0 ? sc2_pol().wgrad_wgs : wg_auto;
    long long chunks = (wg_target + tiles - 1) / tiles;
    long long rows = (M + chunks - 1) / chunks;
    rows = (rows + WG_SLAB - 1) / WG_SLAB * WG_SLAB;
    if (rows < 8 * WG_SLAB) rows = 8 * WG_SLAB;
    chunks = (M + rows - 1) / rows;
    chunks = (chunks + 7) / 8 * 8;      // the kernel deals the ranges to the eight XCDs
    a.rows_per_block = (int)rows;
    const long long grid = tiles * chunks;
    SC2_REQUIRE(grid > 0 && grid < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv2d_wgrad: grid out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)a.Cout * a.K * sizeof(float), s);
    SC2_REQUIRE(e == hipSuccess, SC2_ERR_LAUNCH, "conv2d_wgrad: memset failed: %s", hipGetErrorString(e));
    const size_t lds = (size_t)WG_STAGES * 2 * WG_IMG;
    if (a.x_abs) hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3((unsigned)grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3((unsigned)grid), dim3(256), lds, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
