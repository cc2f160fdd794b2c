// Implicit-GEMM convolution for gfx950 (MI355X): bf16 NHWC activations, bf16 packed weights,
// f32 accumulation on v_mfma_f32_16x16x32_bf16.
//
//   C[m, n] = sum_k A[m, k] * B[k, n]
//   m = (img, oh, ow)            M = N*OH*OW
//   n = output channel           (Cout)
//   k = (kh, kw, ci)             K = KH*KW*Cin, gathered from the NHWC input on the fly
//
// One kernel covers every transform on the bottleneck path (reference: nn.Conv2d at
// sc2bench/models/layer.py:475-493) and, as a 1x1 conv on |x| with a multiplicative epilogue,
// CompressAI's GDN1 (layer.py:478,481,488,491).
//
// Structure (per 256-thread workgroup = 4 waves, one BM x BN output tile):
//   * A and B k-slabs (BK = 32) go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no
//     ds_write) into a 4-deep ring of XOR-swizzled LDS images; the LDS image is lane-linear per
//     wave-instruction, so the swizzle is applied to each lane's SOURCE address (which chunk it fetches).
//     Three slabs stay in flight across the single raw s_barrier per slab behind a counted
//     s_waitcnt vmcnt(N); fragments are read with inline-asm ds_read_b128 (a compiler-visible LDS read
//     would make hipcc drain vmcnt(0) in front of it) -- conflict-free per the lane-group table of
//     MI355X_MICROARCH.md section LDS.
//   * 16-byte chunks never straddle a filter tap (Cin % 8 == 0), so each chunk is one lane of a
//     direct-to-LDS load; out-of-image taps and the K tail source a 16-byte zero block instead.
//   * the epilogue stages the f32 accumulators through LDS so that global stores are
//     whole 16-byte channel runs (NHWC) or pixel runs (NCHW) and the fused element-wise
//     epilogues (GDN / IGDN / bias / residual) read their operands coalesced.
//   * workgroup ids are remapped so that workgroups sharing an XCD (id % 8) cover
//     neighbouring tiles (shared input halo and the same weight panel stay in that XCD's L2).
//
// This header holds every template of the implicit-GEMM family; the instantiations are spread over the
// conv_inst_*.hip translation units (explicit instantiation of the launchers) so that they compile in parallel, and
// conv_igemm.hip holds the C-ABI dispatcher with the matching `extern template` declarations.
#pragma once
#include <stdlib.h>

#include "sc2_common.h"

namespace sc2conv {

// epilogues that read an activation operand x through ep_x (GDN forms: y = x * f(beta + acc); residual add)
template <bool SQ = true>
__host__ __device__ __forceinline__ constexpr bool epi_needs_x(int epi) {
    return epi == SC2_EPI_GDN || epi == SC2_EPI_IGDN || epi == SC2_EPI_BIAS_ADD_RELU ||
           (SQ && (epi == SC2_EPI_GDN2 || epi == SC2_EPI_IGDN2));
}

// eight bf16 values squared (f32 product, rounded to nearest even by the pack): the A operand of the squared-form GDN
// GEMM norm^2 = beta + gamma x^2 (CompressAI GDN, bmshj2018_factorized)
__device__ __forceinline__ uint4 bf16x8_square(uint4 v) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float lo = __builtin_bit_cast(float, w[t] << 16), hi = __builtin_bit_cast(float, w[t] & 0xFFFF0000u);
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t sq = {lo * lo, hi * hi};
        w[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(sq, bf16x2_t));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}


struct ConvArgs {
    const uint16_t *__restrict__ x;
    const uint16_t *__restrict__ w;
    void *__restrict__ y;
    const uint16_t *__restrict__ ep_x;
    const float *__restrict__ ep_beta;
    int N, H, W, Cin, Cout;
    int KH, KW, SH, SW, PH, PW;
    int OH, OW, OHW, M;
    int Kpad, KT;
    int n_ntiles;
    int aop, epi, out;
    int g_pitch;   // fused GDN: row pitch (elements) of the packed gamma matrix handed in through ep_x
    int b_kt_stride;    // element stride of the packed weights between k-slabs: 32 (row-major rows of Kpad) or Cout_pad*32
                        // (SC2_K_B_TILE_MAJOR: [k-slab][row][32], a slab's B tile is contiguous: whole 128-byte lines per load)
    int b_row_stride;   // element stride between weight rows: Kpad or 32
    int k_slab_major;   // K ordered (channel slab of 32, tap, channel) instead of (tap, channel): needs Cin % 32 == 0
    int dbg;   // development switches (SC2_CONV_DEBUG): bit 0 skips the store epilogue, bit 1 the K loop
    unsigned x_bytes, w_bytes;   // sizes of x and of the packed weights when both are < 2 GB (buffer-addressed loads), else 0
    int o_H, o_W, o_sh, o_sw, o_h0, o_w0;   // NHWC output scatter (o_H == 0: dense): pixel (oh, ow) -> (oh*o_sh+o_h0, ..)
    int DH, DW;                             // filter dilation (1 = none); read by the Cfg::DIL instantiations only
    const uint16_t *__restrict__ ep_x2;     // Cfg::BWD: the second per-element operand (the GDN's input x)
    void *__restrict__ y2;                  // Cfg::BWD, PRE epilogues: the second output (the direct term)
};

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, bool STATIC_, int CIN_, int KH_, int KW_, int SH_, int SW_,
          int PH_, int PW_, int STAGES_ = 2, bool EPX_ = false, bool SQ_ = false, bool DIL_ = false, bool BWD_ = false>
struct Cfg {
    // BWD: the GDN1-backward epilogues (SC2_EPI_*GDN1_BWD_*: two per-element operands, two outputs; conv_store_tile_gdn_bwd).
    static constexpr bool BWD = BWD_;
    // DIL: runtime filter dilation (ConvArgs::DH / DW): input pixel of tap (kh, kw) = (oh SH - PH + kh DH, ow SW - PW + kw DW).
    // A compile-time property of its own instantiations (the atrous layers of the dense-prediction heads), like SQ.
    static constexpr bool DIL = DIL_;
    // SQ: the squared-form GDN build (CompressAI GDN: SC2_AOP_SQUARE operand, SC2_EPI_GDN2 / _IGDN2 epilogues).  Its code
    // exists only in the instantiations that set it: in the shared epilogues a runtime case costs every kernel two
    // transcendental ops per element (measured +3 % on the decoder layers), so it is a compile-time property.
    static constexpr bool SQ = SQ_;
    // EPX: prefetch the epilogue operand (GDN's x / the residual) before the K loop (+32 VGPRs); instantiated only
    // for the launches whose epilogue reads one
    static constexpr bool EPX = EPX_;
    static constexpr int BM = BM_, BN = BN_, BK = 32;
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
    static constexpr bool STATIC = STATIC_;
    static constexpr int CIN = CIN_, KH = KH_, KW = KW_, SH = SH_, SW = SW_, PH = PH_, PW = PW_;
    static constexpr int KC = BK / 8;  // 16-byte chunks per tile row
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int MT = WM / 16, NT = WN / 16;
#ifndef SC2_CONV_ORDER
#define SC2_CONV_ORDER 1
#endif
#ifndef SC2_CONV_PRIO
#define SC2_CONV_PRIO 0
#endif
    // LDS ring depth (k-slabs).  Measured on MI355X (tools/layer_times.py): short-K / HBM-bound layers want
    // occupancy (2 slabs -> 4 workgroups per CU), long-K MFMA-bound layers want 3 slabs (3 workgroups per CU).
    static constexpr int STAGES = STAGES_;
    static constexpr int BNP = (BN + 63) / 64 * 64;        // B rows staged (whole 1-KB wave-instructions per wave)
    static constexpr int A_IPW = BM / 64;                  // direct-to-LDS instructions per wave per slab (A)
    static constexpr int B_IPW = BNP / 64;                 //                                              (B)
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BNP * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int STAGE_ROWS = WAVES_M * 16;
    static constexpr int MAIN_LDS = STAGES * STAGE_BYTES;
    static constexpr int EPI_LDS = STAGE_ROWS * (BN + 4) * 4;
    // fused GDN epilogue (tile covers every output channel): |x| image [BM][XC chunks] + gamma image [BN][XC chunks]
    static constexpr int XC = (BN + 31) / 32 * 4;          // 16-byte chunks per row (K of the second GEMM, padded to 32)
    static constexpr int XSW = (XC % 8 == 0) ? 7 : 3;      // chunk XOR mask: conflict-free ds_read_b128 for XC = 8 / 12
    static constexpr int FUSE_LDS = (BM + BN) * XC * 16;
    static constexpr int IMG_LDS = BM * (BN * 2 + 16);     // bf16 store image (ImgPad)
    static constexpr int LDS0 = MAIN_LDS > EPI_LDS ? MAIN_LDS : EPI_LDS;
    static constexpr int LDS1 = LDS0 > FUSE_LDS ? LDS0 : FUSE_LDS;
    static constexpr int LDS2 = LDS1 > IMG_LDS ? LDS1 : IMG_LDS;
    static constexpr int LDS_BYTES = (BWD_ && 2 * IMG_LDS > LDS2) ? 2 * IMG_LDS : LDS2;   // BWD: two operand images side by side
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(BM % (WAVES_M * 16) == 0 && BN % (WAVES_N * 16) == 0, "wave tiling");
    static_assert(BM % 64 == 0, "A rows per wave-instruction");
};

// byte offset of 16-byte chunk `c` of row `r` in a [rows][BK] bf16 LDS tile (BK = 32 -> 64-byte rows).
// chunk index XORed with (r >> 1) & 3: the 16 rows x 4 chunks one MFMA operand read touches
// land on 16 distinct 16-byte slots of the 256-byte bank row in every ds_read_b128 lane group.
__device__ __forceinline__ int lds_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 3)) << 4); }

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

static __device__ uint4 g_zero16;   // (one per translation unit) 16 zero bytes: the source of every out-of-image / K-tail chunk

// Buffer descriptors and buffer-addressed direct-to-LDS loads.  The host pass of hipcc parses kernel bodies too and has
// neither the type nor the builtins: it gets stand-ins (a kernel whose body fails to parse silently loses its host stub).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)bytes, 0x00020000);
}
// 16 bytes per lane from base + voff + soff (voff per lane, out of range -> zeros; soff scalar) to LDS at dst + 16 * lane
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, lds_ptr_t dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, (int)voff, (int)soff, 0, 0);
}
#else
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
#endif

__device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// ------------------------------------------------------------------------------------------------ store epilogue
// Accumulator layout.  The MFMAs are issued with the WEIGHT fragment as the A operand and the activation fragment as
// the B operand, so D = W X^T and lane (frow, fq) of accumulator tile (i, j) holds FOUR CONSECUTIVE CHANNELS
//     out[pixel = wm*WM + i*16 + frow][channel = wn*WN + j*16 + fq*4 + e],  e = 0..3
// i.e. 8 contiguous bytes of the bf16 NHWC output: one ds_write_b64 per accumulator tile instead of four scalar
// writes, and the element-wise epilogues see their per-channel operands as float4.
//
// bf16 NHWC (the format between kernels): ONE pass.  The whole BM x BN tile is transposed through a bf16 LDS image
// (rows padded by 16 B, or chunk-XOR-swizzled for the 8-wave tiles): the epilogue operand x (GDN / residual) is
// parked in the image with coalesced 16-byte accesses, every lane updates its 8-byte slots in place in f32
// (x -> y), and the image is streamed out in whole 16-byte channel runs.  Measured before this layout
// (tools/attic/epi_share.sh): the 4-pass f32 staging cost as much as the whole K loop on the K <= 512 layers.
// f32 outputs (latent, module-level API, fc): MT passes through an f32 staging buffer, as before.
template <class C>
struct ImgPad {   // 4-wave tiles: row pitch BN*2 + 16 bytes (ds_write_b64 of 16 rows: 2-way conflicts at worst)
    static constexpr int PITCH = C::BN * 2 + 16;
    static constexpr int BYTES = C::BM * PITCH;
    static __device__ __forceinline__ int off(int row, int c16) { return row * PITCH + (c16 << 4); }
};
template <class C>
struct ImgXor {   // 8-wave tiles (BN*2/16 >= 16 chunks per row): power-of-two pitch, chunk index XOR (row & 15)
    static constexpr int PITCH = C::BN * 2;
    static constexpr int BYTES = C::BM * PITCH;
    static __device__ __forceinline__ int off(int row, int c16) { return row * PITCH + ((c16 ^ (row & 15)) << 4); }
};

template <class C, int NTHREADS>
struct EpiGeom {
    static constexpr int CPR = C::BN / 8;                                  // 16-byte chunks per tile row
    static constexpr int Q = C::BM * CPR;                                  // chunks per tile
    static constexpr int QPT = (Q + NTHREADS - 1) / NTHREADS;              // chunks per thread
    static constexpr int QPT_PASS = (C::STAGE_ROWS * CPR + NTHREADS - 1) / NTHREADS;   // f32 staging passes
};

// Issues, before the main loop, the loads of the epilogue operand (GDN's x / the residual) this thread will park in
// the LDS image: their latency hides behind the whole K loop.
template <class C, int NTHREADS>
__device__ __forceinline__ void conv_prefetch_epx(const ConvArgs &p, int tid, int m0, int n0,
                                                  uint4 (&epx)[EpiGeom<C, NTHREADS>::QPT]) {
    constexpr int CPR = EpiGeom<C, NTHREADS>::CPR, QPT = EpiGeom<C, NTHREADS>::QPT, Q = EpiGeom<C, NTHREADS>::Q;
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int q = tid + r * NTHREADS;
        const int row = q / CPR, cc = q - row * CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        const bool ok = (q < Q) & (m < p.M) & (n < p.Cout);
        epx[r] = ok ? *reinterpret_cast<const uint4 *>(p.ep_x + (long long)m * p.Cout + n) : make_uint4(0u, 0u, 0u, 0u);
    }
}

__device__ __forceinline__ long long conv_out_offset(const ConvArgs &p, int m, int n, bool &ok) {
    ok = true;
    if (p.o_H > 0) {   // strided scatter (transposed-convolution parity classes of the data gradient)
        const int img = m / p.OHW;
        const int rem = m - img * p.OHW;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int yh = oh * p.o_sh + p.o_h0, yw = ow * p.o_sw + p.o_w0;
        ok = (yh < p.o_H) & (yw < p.o_W);
        return (((long long)img * p.o_H + yh) * p.o_W + yw) * p.Cout + n;
    }
    return (long long)m * p.Cout + n;
}

// bf16 NHWC store through the LDS image `img` (Img::BYTES, idle LDS).  x_in_image: the image already holds the
// epilogue operand x (the fused conv + GDN1 of the 256-wide tile leaves it there).
template <class C, class Img, int NTHREADS, bool PREFETCHED>
__device__ __forceinline__ void conv_store_tile_bf16(const ConvArgs &p, unsigned char *img, f32x4_t (&acc)[C::MT][C::NT],
                                                     int tid, int wm, int wn, int frow, int fq, int m0, int n0, int epi,
                                                     const uint4 *epx, bool x_in_image, int valid_rows = C::BM) {
    constexpr int MT = C::MT, NT = C::NT;
    constexpr int CPR = EpiGeom<C, NTHREADS>::CPR, QPT = EpiGeom<C, NTHREADS>::QPT, Q = EpiGeom<C, NTHREADS>::Q;
    const int Cout = p.Cout;
    const bool needs_x = epi_needs_x<C::SQ>(epi);
    if (needs_x && !x_in_image) {
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const int q = tid + r * NTHREADS;
            const int row = q / CPR, cc = q - row * CPR;
            if (q < Q) {
                uint4 v;
                if (PREFETCHED) {
                    v = epx[r];
                } else {
                    const int m = m0 + row, n = n0 + cc * 8;
                    v = ((m < p.M) & (n < Cout)) ? *reinterpret_cast<const uint4 *>(p.ep_x + (long long)m * Cout + n)
                                                 : make_uint4(0u, 0u, 0u, 0u);
                }
                *reinterpret_cast<uint4 *>(img + Img::off(row, cc)) = v;
            }
        }
        __syncthreads();
    }
    float4 bj[NT];
    if (epi != SC2_EPI_NONE) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * C::WN + j * 16 + fq * 4;
            bj[j] = n < Cout ? *reinterpret_cast<const float4 *>(p.ep_beta + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wm * C::WM + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wn * C::WN + j * 16 + fq * 4;
            unsigned char *slot = img + Img::off(row, col >> 3) + (col & 7) * 2;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (epi != SC2_EPI_NONE) {
                float xv[4] = {0.f, 0.f, 0.f, 0.f};
                if (needs_x) {
                    const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                    xv[0] = __builtin_bit_cast(float, xr.x << 16);
                    xv[1] = __builtin_bit_cast(float, xr.x & 0xFFFF0000u);
                    xv[2] = __builtin_bit_cast(float, xr.y << 16);
                    xv[3] = __builtin_bit_cast(float, xr.y & 0xFFFF0000u);
                }
                const float b[4] = {bj[j].x, bj[j].y, bj[j].z, bj[j].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float norm = b[t] + v[t];
                    float r;
                    if (epi == SC2_EPI_GDN) r = xv[t] * (1.0f / norm);
                    else if (epi == SC2_EPI_IGDN) r = xv[t] * norm;
                    else if (C::SQ && epi == SC2_EPI_GDN2) r = xv[t] * rsqrtf(norm);
                    else if (C::SQ && epi == SC2_EPI_IGDN2) r = xv[t] * sqrtf(norm);
                    else if (epi == SC2_EPI_BIAS) r = norm;
                    else if (epi == SC2_EPI_BIAS_RELU) r = fmaxf(norm, 0.f);
                    else if (epi == SC2_EPI_BIAS_LEAKY_RELU) r = norm > 0.f ? norm : 0.01f * norm;
                    else r = fmaxf(norm + xv[t], 0.f);
                    v[t] = r;
                }
            }
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2 *>(slot) = o;
        }
    }
    __syncthreads();
    uint16_t *y = reinterpret_cast<uint16_t *>(p.y);
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int q = tid + r * NTHREADS;
        const int row = q / CPR, cc = q - row * CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        if ((q < Q) & (m < p.M) & (n < Cout) & (row < valid_rows)) {
            bool ok;
            const long long o = conv_out_offset(p, m, n, ok);
            if (ok) *reinterpret_cast<uint4 *>(y + o) = *reinterpret_cast<const uint4 *>(img + Img::off(row, cc));
        }
    }
}

// GDN1 backward (Cfg::BWD, round 5): the element-wise halves of the backward in the epilogues of its two GEMMs.  Both per-element
// operands are parked in LDS images with coalesced 16-byte accesses (ep_x -> img, ep_x2 -> img2), every lane updates its 8-byte
// slots in place from its accumulators, and the images are streamed out in whole channel runs (img -> y, img2 -> y2).
//   PRE  (acc = gamma |x|, img = g, img2 = x):  n = beta + acc;  GDN: dd = g / n, dn = -dd x / n;  inverse: dd = g n, dn = g x
//        img <- dn (d_norm), img2 <- dd (the direct term of dL/dx)
//   POST (acc = gamma^T d_norm, img = dd, img2 = x):  img <- dd + sign(x) acc  (sign(0) = 0, as torch.abs differentiates)
// Same arithmetic, in f32, as gdn_bwd_pre_kernel / gdn_bwd_post_kernel -- except that n is the f32 accumulator here where the
// two-launch form read it back rounded to bf16.
template <class C, class Img, int NTHREADS>
__device__ __forceinline__ void conv_store_tile_gdn_bwd(const ConvArgs &p, unsigned char *smem, f32x4_t (&acc)[C::MT][C::NT],
                                                        int tid, int wm, int wn, int frow, int fq, int m0, int n0) {
    constexpr int MT = C::MT, NT = C::NT;
    constexpr int CPR = EpiGeom<C, NTHREADS>::CPR, QPT = EpiGeom<C, NTHREADS>::QPT, Q = EpiGeom<C, NTHREADS>::Q;
    unsigned char *img = smem, *img2 = smem + Img::BYTES;
    const int Cout = p.Cout;
    const int epi = p.epi;
    const bool post = epi == SC2_EPI_GDN1_BWD_POST;
    const bool inverse = epi == SC2_EPI_IGDN1_BWD_PRE;
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int q = tid + r * NTHREADS;
        const int row = q / CPR, cc = q - row * CPR;
        if (q < Q) {
            const int m = m0 + row, n = n0 + cc * 8;
            const bool ok = (m < p.M) & (n < Cout);
            const long long o = ok ? (long long)m * Cout + n : 0;
            const uint4 a = *reinterpret_cast<const uint4 *>(p.ep_x + o), b = *reinterpret_cast<const uint4 *>(p.ep_x2 + o);
            *reinterpret_cast<uint4 *>(img + Img::off(row, cc)) = ok ? a : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(img2 + Img::off(row, cc)) = ok ? b : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    float4 bj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * C::WN + j * 16 + fq * 4;
        bj[j] = (!post && n < Cout) ? *reinterpret_cast<const float4 *>(p.ep_beta + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wm * C::WM + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wn * C::WN + j * 16 + fq * 4;
            const int so = Img::off(row, col >> 3) + (col & 7) * 2;
            const uint2 ar = *reinterpret_cast<const uint2 *>(img + so), br = *reinterpret_cast<const uint2 *>(img2 + so);
            const float av[4] = {__builtin_bit_cast(float, ar.x << 16), __builtin_bit_cast(float, ar.x & 0xFFFF0000u),
                                 __builtin_bit_cast(float, ar.y << 16), __builtin_bit_cast(float, ar.y & 0xFFFF0000u)};
            const float xv[4] = {__builtin_bit_cast(float, br.x << 16), __builtin_bit_cast(float, br.x & 0xFFFF0000u),
                                 __builtin_bit_cast(float, br.y << 16), __builtin_bit_cast(float, br.y & 0xFFFF0000u)};
            const float b[4] = {bj[j].x, bj[j].y, bj[j].z, bj[j].w};
            float o1[4], o2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (post) {
                    const float sgn = xv[t] > 0.f ? 1.0f : (xv[t] < 0.f ? -1.0f : 0.f);
                    o1[t] = av[t] + sgn * acc[i][j][t];
                    o2[t] = 0.f;
                } else {
                    const float norm = b[t] + acc[i][j][t];
                    if (inverse) {
                        o1[t] = av[t] * xv[t];
                        o2[t] = av[t] * norm;
                    } else {
                        const float rn = 1.0f / norm;
                        o2[t] = av[t] * rn;
                        o1[t] = -o2[t] * xv[t] * rn;
                    }
                }
            }
            uint2 w1, w2;
            w1.x = pack_bf16x2(o1[0], o1[1]);
            w1.y = pack_bf16x2(o1[2], o1[3]);
            w2.x = pack_bf16x2(o2[0], o2[1]);
            w2.y = pack_bf16x2(o2[2], o2[3]);
            *reinterpret_cast<uint2 *>(img + so) = w1;
            if (!post) *reinterpret_cast<uint2 *>(img2 + so) = w2;
        }
    }
    __syncthreads();
    uint16_t *y = reinterpret_cast<uint16_t *>(p.y), *y2 = reinterpret_cast<uint16_t *>(p.y2);
#pragma unroll
    for (int r = 0; r < QPT; ++r) {
        const int q = tid + r * NTHREADS;
        const int row = q / CPR, cc = q - row * CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        if ((q < Q) & (m < p.M) & (n < Cout)) {
            const long long o = (long long)m * Cout + n;
            *reinterpret_cast<uint4 *>(y + o) = *reinterpret_cast<const uint4 *>(img + Img::off(row, cc));
            if (!post) *reinterpret_cast<uint4 *>(y2 + o) = *reinterpret_cast<const uint4 *>(img2 + Img::off(row, cc));
        }
    }
}

// f32 outputs: MT passes, pass i stages tile-row i of every wave (WAVES_M*16 rows x BN cols, f32) through LDS so
// that global stores are whole channel runs (NHWC) or pixel runs (NCHW), element-wise epilogue applied on the way
// out.  x_img != nullptr: the GDN operand x is an LDS-resident bf16 image [tile row][512 B], chunk XOR (row & 15)
// (fused conv + IGDN of the 256-wide tile); `smem` is then the staging area with unpadded rows (rs_override).
template <class C, int NTHREADS>
__device__ __forceinline__ void conv_store_tile_f32(const ConvArgs &p, unsigned char *smem, f32x4_t (&acc)[C::MT][C::NT],
                                                    int tid, int wm, int wn, int frow, int fq, int m0, int n0, int epi,
                                                    const unsigned char *x_img = nullptr, int rs_override = 0) {
    constexpr int BN = C::BN, MT = C::MT, NT = C::NT;
    float *stage = reinterpret_cast<float *>(smem);
    // symbol output: tiles narrower than 128 channels only (the last encoder conv), decided at compile time so that the
    // wide kernels' code is what it was (a runtime test here sent the 256-wide kernel's accumulators to scratch)
    constexpr bool SYM_OK = C::BN < 128;
    const bool sym = SYM_OK && p.out == SC2_OUT_I32_NCHW_SYM;
    const bool nchw = SYM_OK ? (p.out == SC2_OUT_F32_NCHW || sym) : (p.out == SC2_OUT_F32_NCHW);
    const int RS = rs_override ? rs_override : (nchw ? BN + 1 : BN + 4);  // row stride in floats (bank spread)
    const int Cout = p.Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sr = wm * 16 + frow;
                const int sc = wn * C::WN + j * 16 + fq * 4 + e;
                stage[sr * RS + sc] = acc[i][j][e];
            }
        }
        __syncthreads();
        if (!nchw) {
            constexpr int CPR = BN / 8;  // 8-channel chunks per row
#pragma unroll
            for (int rq = 0; rq < EpiGeom<C, NTHREADS>::QPT_PASS; ++rq) {
                const int q = tid + rq * NTHREADS;
                if (q >= C::STAGE_ROWS * CPR) continue;
                const int sr = q / CPR, cc = q - sr * CPR;
                const int m = m0 + (sr >> 4) * C::WM + i * 16 + (sr & 15);
                const int n = n0 + cc * 8;
                if (m >= p.M || n >= Cout) continue;
                float v[8];
                {
                    const float4 v0 = *reinterpret_cast<const float4 *>(stage + sr * RS + cc * 8);
                    const float4 v1 = *reinterpret_cast<const float4 *>(stage + sr * RS + cc * 8 + 4);
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
                    v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                }
                bool ok;
                const long long o = conv_out_offset(p, m, n, ok);
                if (!ok) continue;
                if (epi != SC2_EPI_NONE) {
                    float b[8];
                    {
                        const float4 b0 = *reinterpret_cast<const float4 *>(p.ep_beta + n);
                        const float4 b1 = *reinterpret_cast<const float4 *>(p.ep_beta + n + 4);
                        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
                        b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
                    }
                    float xv[8];
                    if (epi_needs_x<C::SQ>(epi)) {
                        uint4 xr;
                        if (x_img) {
                            const int trow = (sr >> 4) * C::WM + i * 16 + (sr & 15);
                            xr = *reinterpret_cast<const uint4 *>(x_img + trow * 512 + ((cc ^ (trow & 15)) << 4));
                        } else {
                            xr = *reinterpret_cast<const uint4 *>(p.ep_x + (long long)m * Cout + n);   // always dense
                        }
                        const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            xv[2 * t] = __builtin_bit_cast(float, xw[t] << 16);
                            xv[2 * t + 1] = __builtin_bit_cast(float, xw[t] & 0xFFFF0000u);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; ++t) xv[t] = 0.f;
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float norm = b[t] + v[t];
                        float r;
                        if (epi == SC2_EPI_GDN) r = xv[t] * (1.0f / norm);
                        else if (epi == SC2_EPI_IGDN) r = xv[t] * norm;
                        else if (C::SQ && epi == SC2_EPI_GDN2) r = xv[t] * rsqrtf(norm);
                        else if (C::SQ && epi == SC2_EPI_IGDN2) r = xv[t] * sqrtf(norm);
                        else if (epi == SC2_EPI_BIAS) r = norm;
                        else if (epi == SC2_EPI_BIAS_RELU) r = fmaxf(norm, 0.f);
                    else if (epi == SC2_EPI_BIAS_LEAKY_RELU) r = norm > 0.f ? norm : 0.01f * norm;
                        else r = fmaxf(norm + xv[t], 0.f);
                        v[t] = r;
                    }
                }
                float *yo = reinterpret_cast<float *>(p.y) + o;
                *reinterpret_cast<float4 *>(yo) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {
            // f32 NCHW: lanes run along pixels so each channel plane gets contiguous runs.
            for (int q = tid; q < C::STAGE_ROWS * BN; q += NTHREADS) {
                const int cidx = q / C::STAGE_ROWS, sr = q - cidx * C::STAGE_ROWS;
                const int m = m0 + (sr >> 4) * C::WM + i * 16 + (sr & 15);
                const int n = n0 + cidx;
                if (m >= p.M || n >= Cout) continue;
                float v = stage[sr * RS + cidx];
                if (epi != SC2_EPI_NONE) {
                    const float norm = p.ep_beta[n] + v;
                    float xv = 0.f;
                    if (epi_needs_x<C::SQ>(epi)) {
                        if (x_img) {
                            const int trow = (sr >> 4) * C::WM + i * 16 + (sr & 15);
                            xv = bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(
                                x_img + trow * 512 + (((cidx >> 3) ^ (trow & 15)) << 4) + (cidx & 7) * 2));
                        } else {
                            xv = bf16_bits_to_f32(p.ep_x[(long long)m * Cout + n]);
                        }
                    }
                    if (epi == SC2_EPI_GDN) v = xv * (1.0f / norm);
                    else if (epi == SC2_EPI_IGDN) v = xv * norm;
                    else if (C::SQ && epi == SC2_EPI_GDN2) v = xv * rsqrtf(norm);
                    else if (C::SQ && epi == SC2_EPI_IGDN2) v = xv * sqrtf(norm);
                    else if (epi == SC2_EPI_BIAS) v = norm;
                    else if (epi == SC2_EPI_BIAS_RELU) v = fmaxf(norm, 0.f);
                    else if (epi == SC2_EPI_BIAS_LEAKY_RELU) v = norm > 0.f ? norm : 0.01f * norm;
                    else v = fmaxf(norm + xv, 0.f);
                }
                const int img = m / p.OHW;
                const int pix = m - img * p.OHW;
                if constexpr (SYM_OK) {
                    if (sym) {   // quantised straight from the accumulator: round-half-even(v - median), as eb_symbols
                        reinterpret_cast<int32_t *>(p.y)[((long long)img * Cout + n) * p.OHW + pix] = (int32_t)rintf(v - p.ep_beta[n]);
                        continue;
                    }
                }
                reinterpret_cast<float *>(p.y)[((long long)img * Cout + n) * p.OHW + pix] = v;
            }
        }
    }
}

// Fused GDN1 / inverse GDN1 of a tile that holds every output channel of its pixels (BN <= 96): norm = beta + gamma |x|
// is a second, LDS-resident GEMM - |x| (bf16) is written to an LDS image straight from the accumulators, gamma is staged
// next to it, and y = x / norm (or x * norm) is applied to the f32 accumulators: the GDN costs no HBM traffic.
// Call with the LDS idle; returns with the LDS idle.
template <class C>
__device__ __forceinline__ void conv_fused_gdn_small(const ConvArgs &p, unsigned char *smem, f32x4_t (&acc)[C::MT][C::NT],
                                                     int tid, int wm, int wn, int frow, int fq, int n0) {
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;

        constexpr int XC = C::XC, XSW = C::XSW, ROWB = XC * 16;
        unsigned char *Xi = smem;
        unsigned char *Gi = smem + BM * ROWB;
        if (XC * 8 > BN) {   // K padding of the second GEMM: zero the chunks past the last channel
            constexpr int PADC = XC - BN / 8;
            for (int q = tid; q < BM * PADC; q += 256) {
                const int r = q / PADC, c = BN / 8 + (q - r * PADC);
                *reinterpret_cast<uint4 *>(Xi + r * ROWB + ((c ^ ((r >> 1) & XSW)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wm * C::WM + i * 16 + frow;
                const int col = wn * C::WN + j * 16 + fq * 4;   // 4 consecutive channels of pixel r
                uint2 h;                                        // |x| as bf16
                h.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]) & 0x7FFF7FFFu;
                h.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]) & 0x7FFF7FFFu;
                *reinterpret_cast<uint2 *>(Xi + r * ROWB + (((col >> 3) ^ ((r >> 1) & XSW)) << 4) + (col & 7) * 2) = h;
            }
        const uint16_t *gamma = p.ep_x;   // packed bf16 [rows >= BN][g_pitch], zero padded
        for (int q = tid; q < BN * XC; q += 256) {
            const int r = q / XC, c = q - r * XC;
            const uint4 v = *reinterpret_cast<const uint4 *>(gamma + (long long)r * p.g_pitch + c * 8);
            *reinterpret_cast<uint4 *>(Gi + r * ROWB + ((c ^ ((r >> 1) & XSW)) << 4)) = v;
        }
        __syncthreads();
        f32x4_t nrm[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) nrm[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < XC / 4; ++ks) {
            bf16x8_t xa[MT], gb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * C::WM + i * 16 + frow;
                xa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(
                                                         Xi + r * ROWB + (((4 * ks + fq) ^ ((r >> 1) & XSW)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * C::WN + j * 16 + frow;
                gb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(
                                                         Gi + r * ROWB + (((4 * ks + fq) ^ ((r >> 1) & XSW)) << 4)));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    nrm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[j], xa[i], nrm[i][j], 0, 0, 0);
        }
        const bool inverse = p.epi == SC2_EPI_FUSED_IGDN;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * C::WN + j * 16 + fq * 4;
            const float4 b4 = col < p.Cout ? *reinterpret_cast<const float4 *>(p.ep_beta + col)
                                           : make_float4(1.f, 1.f, 1.f, 1.f);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float norm = b[e] + nrm[i][j][e];
                    acc[i][j][e] = inverse ? acc[i][j][e] * norm : acc[i][j][e] * (1.0f / norm);
                }
        }
        __syncthreads();   // the images are dead; the staging buffer below reuses their LDS
    }

template <class C>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int BM = C::BM, BN = C::BN, KC = C::KC;
    constexpr int MT = C::MT, NT = C::NT, S = C::STAGES;
    constexpr int A_IPW = C::A_IPW, B_IPW = C::B_IPW, L = A_IPW + B_IPW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const int Cin = C::STATIC ? C::CIN : p.Cin;
    const int KH = C::STATIC ? C::KH : p.KH;
    const int KW = C::STATIC ? C::KW : p.KW;
    const int SH = C::STATIC ? C::SH : p.SH;
    const int SW = C::STATIC ? C::SW : p.SW;
    const int PH = C::STATIC ? C::PH : p.PH;
    const int PW = C::STATIC ? C::PW : p.PW;
    const int DH = C::DIL ? p.DH : 1, DW = C::DIL ? p.DW : 1;
    const int CIN8 = Cin >> 3;
    const int H = p.H, W = p.W;

    // --- XCD-aware workgroup remap (bijective form) ---
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int ntile = bid % p.n_ntiles;
    const int mtile = bid / p.n_ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // --- per-lane gather state.  Wave-instruction q = j*4 + wave fills LDS rows [16q, 16q+16) of a slab; lane l
    //     owns 16-byte position 64q + l = row 16q + (l >> 2), stored chunk l & 3, which under the read swizzle
    //     holds k-chunk (l & 3) ^ ((row >> 1) & 3) = (l & 3) ^ ((l >> 3) & 3): the same for every row of a lane.
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);
    const long long zero_off = zero - p.x, zero_off_w = zero - p.w;   // element offsets of the zero block
    long long a_off[A_IPW];
    int a_ih0[A_IPW], a_iw0[A_IPW];
    bool a_ok[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; ++j) {
        const int m = m0 + (j * 4 + wave) * 16 + (lane >> 2);
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int img = mm / p.OHW;
        const int rem = mm - img * p.OHW;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_ih0[j] = oh * SH - PH;
        a_iw0[j] = ow * SW - PW;
        a_off[j] = ((long long)(img * H + a_ih0[j]) * W + a_iw0[j]) * Cin;
    }
    long long b_off[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j) {
        int rowb = (j * 4 + wave) * 16 + (lane >> 2);
        if (rowb >= BN) rowb = 0;   // padding rows of the staged image: any valid source, never read back
        b_off[j] = (long long)(n0 + rowb) * p.b_row_stride + kc * 8;
    }
    // k state of this lane's chunk column: (kh, kw, c8); advancing by one slab wraps at most WRAPS times
    constexpr int WRAPS = C::STATIC ? (KC + (C::CIN / 8) - 1) / (C::CIN / 8 > 0 ? C::CIN / 8 : 1) : KC;
    int c8 = kc, kh = 0, kw = 0;
    auto wrap_k = [&]() {
#pragma unroll
        for (int rep = 0; rep < WRAPS; ++rep) {
            const bool w1 = c8 >= CIN8;
            c8 -= w1 ? CIN8 : 0;
            kw += w1 ? 1 : 0;
            const bool w2 = kw == KW;
            kw = w2 ? 0 : kw;
            kh += w2 ? 1 : 0;
        }
    };
    if (!(C::STATIC ? (C::CIN % 32 == 0) : (Cin % 32 == 0))) wrap_k();
    const int KT = (p.dbg & 2) ? 0 : p.KT;

    // When Cin % 32 == 0 a slab never straddles a filter tap: tap and channel base are wave-uniform closed forms of
    // the slab index (scalar registers).  With k_slab_major the K axis runs (channel slab, tap, channel): the taps of
    // one 32-channel slab are consecutive slabs, so the overlapping pixels they re-read are still in L1 / L2.
    const bool aligned = C::STATIC ? (C::CIN % 32 == 0) : (Cin % 32 == 0);
    const int spt = aligned ? (CIN8 >> 2) : 1;     // slabs per tap
    const int ntaps = KH * KW;
    auto issue_tile = [&](int kt, int buf) {
        unsigned char *Ab = smem + buf * C::STAGE_BYTES;
        unsigned char *Bb = Ab + C::A_BYTES;
        int t_kh, t_kw;
        long long tap_off;
        bool tap_ok;
        if (aligned) {
            int tap, cb;
            if (p.k_slab_major) { cb = kt / ntaps; tap = kt - cb * ntaps; }
            else { tap = kt / spt; cb = kt - tap * spt; }
            t_kh = (tap / KW) * DH;         // (input-row / -column offset of the tap: tap index x dilation)
            t_kw = (tap - (tap / KW) * KW) * DW;
            tap_off = ((long long)t_kh * W + t_kw) * Cin + cb * 32 + kc * 8;
            tap_ok = kt < KT;
        } else {
            t_kh = kh * DH;
            t_kw = kw * DW;
            tap_off = ((long long)t_kh * W + t_kw) * Cin + c8 * 8;
            tap_ok = kh < KH;   // false for the K tail and for the dummy slabs past KT
        }
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int ih = a_ih0[j] + t_kh, iw = a_iw0[j] + t_kw;
            // bitwise (not short-circuit) so this stays a chain of VALU compares + selects: a conditional branch
            // costs more than the whole address computation (tools/micro/chain2.hip)
            const bool ok = a_ok[j] & tap_ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            const long long off = ok ? a_off[j] + tap_off : zero_off;
            const uint16_t *src = p.x + off;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Ab + (j * 4 + wave) * 1024), 16, 0, 0);
        }
        const bool kt_ok = kt < KT;
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const long long off = kt_ok ? b_off[j] + (long long)kt * p.b_kt_stride : zero_off_w;
            const uint16_t *src = p.w + off;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Bb + (j * 4 + wave) * 1024), 16, 0, 0);
        }
        if (!aligned) {
            c8 += KC;
            wrap_k();
        }
    };

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const uint32_t amask = p.aop == SC2_AOP_ABS ? 0x7FFF7FFFu : 0xFFFFFFFFu;
    const int frow = lane & 15, fq = lane >> 4;
    uint32_t a_rd[MT], b_rd[NT];   // fragment read offsets inside a slab (fixed per lane)
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rd[i] = (uint32_t)lds_off(wm * C::WM + i * 16 + frow, fq);
#pragma unroll
    for (int j = 0; j < NT; ++j) b_rd[j] = (uint32_t)(C::A_BYTES + lds_off(wn * C::WN + j * 16 + frow, fq));

    // epilogue operand (GDN's x / residual), fetched now so that its latency hides behind the K loop
    uint4 epx[C::EPX ? EpiGeom<C, 256>::QPT : 1];
    if constexpr (C::EPX) conv_prefetch_epx<C, 256>(p, tid, m0, n0, epx);

#pragma unroll
    for (int st = 0; st < S - 1; ++st) issue_tile(st, st);

    for (int kt = 0; kt < KT; ++kt) {
        // slab kt has landed once at most (S-2) younger slabs of this wave are outstanding ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");
        // ... and, after the barrier, for every wave; the barrier also frees the slab computed last iteration
        __builtin_amdgcn_s_barrier();
#if SC2_CONV_ORDER == 0
        issue_tile(kt + S - 1, (kt + S - 1) % S);
#endif
        const uint32_t sb = lds_base + (uint32_t)((kt % S) * C::STAGE_BYTES);
        uint4 av[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = lds_read16(sb + a_rd[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = lds_read16(sb + b_rd[j]);
#if SC2_CONV_ORDER == 1
        // the next slab's address arithmetic and direct-to-LDS loads are issued under the fragment reads' latency
        issue_tile(kt + S - 1, (kt + S - 1) % S);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#if SC2_CONV_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        bf16x8_t af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            uint4 v = av[i];
            v.x &= amask; v.y &= amask; v.z &= amask; v.w &= amask;
            if constexpr (C::SQ) {
                if (p.aop == SC2_AOP_SQUARE) v = bf16x8_square(v);   // wave-uniform
            }
            af[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, bv[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D = W X^T
#if SC2_CONV_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    // drain the dummy slabs and make sure every wave is done reading before the epilogue reuses the LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ------------------------------------------------------------------ fused GDN1 / inverse GDN1
    if constexpr (BN <= 96)
        if (p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN)
            conv_fused_gdn_small<C>(p, smem, acc, tid, wm, wn, frow, fq, n0);
    const int epi = (p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN) ? (int)SC2_EPI_NONE : p.epi;
    if (p.dbg & 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }

    if constexpr (C::BWD) {
        conv_store_tile_gdn_bwd<C, ImgPad<C>, 256>(p, smem, acc, tid, wm, wn, frow, fq, m0, n0);
        return;
    }
    if (p.out == SC2_OUT_BF16_NHWC)
        conv_store_tile_bf16<C, ImgPad<C>, 256, C::EPX>(p, smem, acc, tid, wm, wn, frow, fq, m0, n0, epi, epx, false);
    else
        conv_store_tile_f32<C, 256>(p, smem, acc, tid, wm, wn, frow, fq, m0, n0, epi);
}

// ======================================================================================================
// 5x5 stride-2 convolution from an LDS-resident input PATCH (second encoder conv, 96 -> 48, layer.py:479-480).
// The generic kernel gathers every (tap, channel-slab) A tile from L2 again: 25 taps re-read each input pixel 6.25
// times, the re-reads miss the 4 MB L2 (PMC: 3.5 GB fetched for 0.62 GB of input) and the launch runs at the
// Infinity Cache's rate.  Here a workgroup owns TWO output rows of one image (<= 128 pixels); per 32-channel slab it
// loads the 7 input rows it needs ONCE (direct-to-LDS, 1.75x the input instead of 6.25x) and builds all 25 taps'
// A fragments from LDS with per-lane addresses.  Patch layout: [column parity][input row][half column j][64 B]: a
// tap reads one parity plane at consecutive j for consecutive output pixels, so the 16-byte-chunk XOR swizzle of the
// slab tiles ((j >> 1) & 3) keeps the fragment reads conflict-free.  Weights come FRAGMENT-MAJOR
// (SC2_K_B_FRAG_MAJOR: [k-step][16-row tile][lane][8 k], 1 KB contiguous per operand fragment) straight from L2 into
// registers, two steps ahead: no barrier inside a channel slab.  52 KB of LDS -> three workgroups per CU, one's
// patch load overlaps the others' MFMAs.  Epilogue (fused GDN1, bf16 store) shared with conv_igemm_kernel.
template <class C>
__global__ __launch_bounds__(256, 3) void conv5s2_patch_kernel(const ConvArgs p) {
    constexpr int BN = C::BN, MT = C::MT, NT = C::NT;
    constexpr int CIN = C::CIN, NCB = CIN / 32, NTAP = 25;
    static_assert(C::STATIC && C::WAVES_N == 1 && C::KH == 5 && C::KW == 5 && C::SH == 2 && CIN % 32 == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave, wn = 0;
    const int frow = lane & 15, fq = lane >> 4;
    const int H = p.H, W = p.W, OW = p.OW, J = OW + 2;
    const int rp_per_img = (p.OH + 1) >> 1;
    const int img = blockIdx.x / rp_per_img;
    const int oh0 = (blockIdx.x - img * rp_per_img) * 2;
    const int valid = (p.OH - oh0 >= 2 ? 2 : 1) * OW;   // real output pixels of this tile (rows beyond are dummies)
    const int m0 = (img * p.OH + oh0) * OW;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);

    // ---- this thread's share of the patch fill: LDS chunk P = tid + 256 k  <->  (plane, row r, half column j, chunk)
    const int n_chunks = 2 * 7 * J * 4;
    constexpr int MAXQ = (2 * 7 * 66 * 4 + 255) / 256;   // J <= 66
    const int nq = (n_chunks + 255) >> 8;
    auto patch_src = [&](int k) {   // element offset inside the image of LDS chunk tid + 256 k, -1: zero block
        const int P = tid + 256 * k;
        const int cphys = P & 3, t = P >> 2;
        const int t2 = t / J, j = t - t2 * J;
        const int plane = t2 / 7, r = t2 - plane * 7;
        const int chunk = cphys ^ ((j >> 1) & 3);
        const int ih = 2 * oh0 - C::PH + r, iw = 2 * j + plane - C::PW;
        const bool ok = (P < n_chunks) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
        return ok ? (ih * W + iw) * CIN + chunk * 8 : -1;
    };
    const uint16_t *ximg = p.x + (long long)img * H * W * CIN;

    // ---- K is split over the four waves BY TAP (wave w takes taps w, w + 4, ...): every wave then streams only its
    //      own quarter of the weights (an M split made each wave stream all 225 KB per tile: 6.5 GB through the vector
    //      L1 per launch, and the kernel ran at that rate), while all waves read the same A fragments from the patch.
    //      Each wave accumulates partial sums for the WHOLE 128 x 48 tile; they are added up through LDS at the end.
    constexpr int MA = C::BM / 16;   // 8 m-tiles
    const uint4 *bfrag = reinterpret_cast<const uint4 *>(p.w) + lane;   // [(step * NT + j) * 64]

    f32x4_t part[MA][NT];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) part[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int NQ = (NTAP + 3) / 4;   // taps per wave (the last one only for wave 0)
#pragma unroll 1
    for (int cb = 0; cb < ((p.dbg & 2) ? 0 : NCB); ++cb) {   // (SC2_CONV_DEBUG bit 1: no K loop)
        if (cb > 0) __syncthreads();   // every wave is done with the previous slab's patch
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            if (k < nq) {
                const int so = patch_src(k);
                const uint16_t *src = so >= 0 ? ximg + so + cb * 32 : zero;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + (wave * 64 + 256 * k) * 16), 16, 0, 0);
            }
        }
        uint4 bq[2][NT];   // this wave's weight fragments, one tap ahead
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[0][j] = bfrag[((cb * NTAP + wave) * NT + j) * 64];
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT) : "memory");   // the patch share has landed (the weights may still fly)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int tap = wave + 4 * q;
            if (tap < NTAP) {   // wave-uniform
                if (tap + 4 < NTAP) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bq[(q + 1) & 1][j] = bfrag[((cb * NTAP + tap + 4) * NT + j) * 64];
                }
                const int kh = tap / 5, kw = tap - kh * 5;
                const int d = kw >> 1;
                const int tap_off = (((kw & 1) * 7 + kh) * J + d) * 64;
                int fr = frow;
                asm volatile("" : "+v"(fr));   // keeps the address arithmetic below INSIDE the tap loop
#pragma unroll
                for (int i = 0; i < MA; ++i) {
                    // pixel of (m-tile i, lane row) -> patch offset (recomputed: registers are what limits occupancy)
                    const int pix = i * 16 + fr;
                    const bool real = pix < 2 * OW;
                    const int orow = real ? (pix >= OW ? 1 : 0) : 1;
                    const int ocol = real ? pix - orow * OW : OW - 1;
                    const int swz = (fq ^ (((ocol + d) >> 1) & 3)) << 4;
                    const bf16x8_t af = __builtin_bit_cast(
                        bf16x8_t, *reinterpret_cast<const uint4 *>(smem + ((2 * orow) * J + ocol) * 64 + tap_off + swz));
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[q & 1][j]), af,
                                                                             part[i][j], 0, 0, 0);
                }
            }
        }
    }
    if (p.dbg & 1) {   // SC2_CONV_DEBUG bit 0: no reduction / epilogue
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(part[i][j]));
        return;
    }
    // ---- add the four waves' partial tiles through an f32 LDS buffer [128][BN]; every wave then takes back the rows
    //      it owns in the epilogue's tiling (wave w: rows 32w .. 32w + 31)
    __syncthreads();   // the patch is dead
    // (one wave at a time: three waves adding concurrently with ds_add_f32 into the same slots ran 2x slower)
    float *red = reinterpret_cast<float *>(smem);
#pragma unroll 1
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < MA; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float4 *slot = reinterpret_cast<float4 *>(red + (i * 16 + frow) * BN + j * 16 + fq * 4);
                    float4 v = make_float4(part[i][j][0], part[i][j][1], part[i][j][2], part[i][j][3]);
                    if (turn > 0) {
                        const float4 o = *slot;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *slot = v;
                }
        }
        __syncthreads();
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(red + (wm * C::WM + i * 16 + frow) * BN + j * 16 + fq * 4);
            acc[i][j] = f32x4_t{v.x, v.y, v.z, v.w};
        }
    __syncthreads();   // the reduction buffer is dead: its LDS becomes the GDN images / the store image
    if constexpr (BN <= 96)
        if (p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN)
            conv_fused_gdn_small<C>(p, smem, acc, tid, wm, wn, frow, fq, 0);
    const int epi = (p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN) ? (int)SC2_EPI_NONE : p.epi;
    conv_store_tile_bf16<C, ImgPad<C>, 256, false>(p, smem, acc, tid, wm, wn, frow, fq, m0, 0, epi, nullptr, false, valid);
}

// Everything behind the K loop of the big-tile kernels: the fused conv + (I)GDN1 second GEMM of the 256-wide tile,
// then the store epilogue.  Shared by the 8-wave and the 4-wave kernels (NTHREADS = 512 / 256).
template <class C, int NTHREADS>
__device__ __forceinline__ void conv_big_epilogue(const ConvArgs &p, unsigned char *smem, f32x4_t (&acc)[C::MT][C::NT], int tid,
                                                  int lane, int wm, int wn, int frow, int fq, int m0, int n0) {
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
    // ---- conv followed by GDN1 / inverse GDN1 in the same launch (the tile holds all 256 channels of its pixels):
    // x goes to an LDS image as bf16, norm = gamma |x| is a second MFMA GEMM whose A operand is that image (|.| on the
    // fragment) and whose B operand, gamma, comes fragment-major from L2; y = x * (beta + norm) (or x / ...) is applied
    // in the store pass with x read back from the image.  No HBM traffic for the GDN.
    if constexpr (BN == 256) {
        if (p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN) {
            unsigned char *Xi = smem;
            unsigned char *ring = smem + BM * 512;   // f32-output staging (BM == 256 only)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int r = wm * C::WM + i * 16 + frow;
                    const int col = wn * C::WN + j * 16 + fq * 4;
                    uint2 h;
                    h.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                    h.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                    *reinterpret_cast<uint2 *>(Xi + r * 512 + (((col >> 3) ^ (r & 15)) << 4) + (col & 7) * 2) = h;
                    acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            // gamma comes FRAGMENT-MAJOR ([16-channel tile][32-deep step][lane][8 k]: one operand fragment = 1 KB
            // contiguous, hip.pack_gamma_fragments) straight from L2 into registers, one step ahead: no LDS ring and no
            // barrier inside the loop, the two waves of a SIMD drift apart and hide each other's waits.
            const uint4 *gfrag = reinterpret_cast<const uint4 *>(p.ep_x) + (long long)(wn * NT) * (BN / 32) * 64 + lane;
            constexpr int NS = BN / 32;
            uint4 gbuf[2][NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) gbuf[0][j] = gfrag[(j * NS + 0) * 64];
            __builtin_amdgcn_s_barrier();   // the x image is complete
            const unsigned char *xrow = Xi + (wm * C::WM + frow) * 512;
#pragma unroll
            for (int ks2 = 0; ks2 < NS; ++ks2) {
                if (ks2 + 1 < NS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) gbuf[(ks2 + 1) & 1][j] = gfrag[(j * NS + ks2 + 1) * 64];
                }
                const int xc = ((4 * ks2 + fq) ^ frow) << 4;   // row & 15 == frow for every fragment row of this lane
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    uint4 v = *reinterpret_cast<const uint4 *>(xrow + i * 16 * 512 + xc);
                    v.x &= 0x7FFF7FFFu; v.y &= 0x7FFF7FFFu; v.z &= 0x7FFF7FFFu; v.w &= 0x7FFF7FFFu;
                    const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, gbuf[ks2 & 1][j]), af, acc[i][j], 0, 0, 0);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_barrier();   // every wave is done with its x-image fragments
            const int epi2 = p.epi == SC2_EPI_FUSED_IGDN ? (int)SC2_EPI_IGDN : (int)SC2_EPI_GDN;
            if (p.out == SC2_OUT_BF16_NHWC)   // x -> y in place in the image, then streamed out
                conv_store_tile_bf16<C, ImgXor<C>, NTHREADS, false>(p, Xi, acc, tid, wm, wn, frow, fq, m0, n0, epi2, nullptr,
                                                               true);
            else if (NTHREADS != 256)         // the ring becomes the f32 staging area, x read back from the image
                conv_store_tile_f32<C, NTHREADS>(p, ring, acc, tid, wm, wn, frow, fq, m0, n0, epi2, Xi, BN);
            return;
        }
    }
    if (p.out == SC2_OUT_BF16_NHWC)
        conv_store_tile_bf16<C, ImgXor<C>, NTHREADS, false>(p, smem, acc, tid, wm, wn, frow, fq, m0, n0, p.epi, nullptr, false);
    else if (NTHREADS != 256)   // (the 4-wave kernel is dispatched for bf16 NHWC outputs only: 256 accumulators + staging spill)
        conv_store_tile_f32<C, NTHREADS>(p, smem, acc, tid, wm, wn, frow, fq, m0, n0, p.epi);
}

// ======================================================================================================
// Big-tile variant for the MFMA-bound layers (Cout % 128 == 0, long K): 512 threads = 8 waves, 256 x BN tile,
// BK = 32 slabs in a 4-deep direct-to-LDS ring.  The 8 waves form two groups of four (wave w and w + 4 share a
// SIMD) that run ONE BARRIER OUT OF STEP: between two barriers one group issues its fragment reads and the next
// slab's direct-to-LDS loads while the other group issues 16 MFMAs, then they swap.  The SIMD's matrix pipe is
// therefore fed by one wave while its partner does the LDS / address work, instead of both stalling together.
//   per wave and slab: PHASES phases of {4 A-fragment reads (+ NT B reads in phase 0), a share of slab t+3's
//   loads, lgkmcnt(0) | barrier | 16 MFMAs | barrier}; a slab's loads are retired with a counted vmcnt one slab
//   before its first read, by every wave, ahead of the barrier that opens that read (RAW); a stage is re-filled only
//   after a barrier that follows the lgkmcnt(0) of its last readers (WAR).
template <int BN_, int WAVES_M_, int WAVES_N_, bool STATIC_, int CIN_, int KH_, int KW_, int SH_, int SW_, int PH_,
          int PW_, int BM_ = 256, int STAGES_ = 4, bool PATCH3_ = false, int PATCH_EXTRA_ = 64>
struct Cfg8 {
    // PATCH3 (window staging): stride-1 convolutions with static KH x KW and padding, runtime Cin % 32 == 0, slab-major
    // K.  Along the flattened NHW pixel index the input pixel of output pixel m at tap (kh, kw) is
    //     g(m) + kh W + kw,   g(m) = img HW + (oh - PH) W + (ow - PW),
    // and g(m) - m changes only at output-row and image boundaries (not at all when OH x OW == H x W).  The pixel
    // operand of all KH KW taps of a 32-channel slab is therefore one window of BM + PATCH_EXTRA consecutive input
    // pixels: it is staged ONCE per slab (20-28 KB instead of KH KW im2col slabs of 16 KB through the L2 -> LDS
    // path, which is what bounds these layers) and every tap reads it at a row shift of kh W + kw, out-of-image taps
    // redirected to a zero row.  The dispatcher checks that the window of every tile fits (sc2_conv2d_fwd).
    static constexpr bool PATCH3 = PATCH3_;
    static constexpr bool SQ = false;
    static constexpr int PATCH_ROWS = BM_ + PATCH_EXTRA_, PATCH_BYTES = PATCH_ROWS * 64;
    static constexpr int PATCH_ZERO = 2 * PATCH_BYTES;          // 64 zero bytes behind the two patch buffers
    static constexpr int PATCH_B0 = PATCH_ZERO + 64;            // weight-slab ring
    // BM = 128 with a 3-deep ring: 72 KB of LDS and <= 128 registers, so TWO workgroups share a CU and one's store
    // epilogue (x image, second GEMM, read-out: ~20 % of a fused conv + IGDN tile) overlaps the other's K loop
    static constexpr int BM = BM_, BN = BN_, BK = 32, KC = 4;
    static constexpr int MIN_WAVES = (BM_ == 128 || (PATCH3_ && BN_ == 128)) ? 4 : 2;   // waves per SIMD the register allocation must allow
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
    static constexpr bool STATIC = STATIC_;
    static constexpr int CIN = CIN_, KH = KH_, KW = KW_, SH = SH_, SW = SW_, PH = PH_, PW = PW_;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int MT = WM / 16, NT = WN / 16;
    static constexpr int PHASES = MT / 4;
    static constexpr int STAGES = STAGES_;
    static constexpr int A_IPW = BM / 16 / 8, B_IPW = BN / 16 / 8;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int STAGE_ROWS = WAVES_M * 16;
    static constexpr int MAIN_LDS = PATCH3_ ? PATCH_B0 + STAGES * B_BYTES : STAGES * STAGE_BYTES;
    static constexpr int EPI_LDS = STAGE_ROWS * (BN + 4) * 4;
    // fused conv + GDN1 (BN == 256 == Cout): x image 256 x 512 B + a 2 x 16 KB gamma-slab ring / store staging
    // (the f32-output form of the fused epilogue stages through 32 KB behind the image: full-height tile only)
    static constexpr int FUSE_LDS = BN == 256 ? BM * 512 + (BM == 256 ? 32768 : 0) : 0;
    static constexpr int IMG_LDS = BM * BN * 2;            // bf16 store image (ImgXor)
    static constexpr int LDS1 = MAIN_LDS > EPI_LDS ? MAIN_LDS : EPI_LDS;
    static constexpr int LDS2 = LDS1 > FUSE_LDS ? LDS1 : FUSE_LDS;
    static constexpr int LDS_BYTES = LDS2 > IMG_LDS ? LDS2 : IMG_LDS;
    static_assert(WAVES_M * WAVES_N == 8 && NT == 4 && MT % 4 == 0, "8 waves, 16 MFMAs per phase");
    static_assert(BN % 128 == 0, "whole direct-to-LDS instructions per wave");
};

template <class C>
__global__ __launch_bounds__(512, C::MIN_WAVES) void conv_igemm8_kernel(const ConvArgs p) {
    constexpr int BM = C::BM, BN = C::BN, KC = C::KC;
    constexpr int MT = C::MT, NT = C::NT, S = C::STAGES, PHASES = C::PHASES;
    constexpr int A_IPW = C::A_IPW, B_IPW = C::B_IPW, L = A_IPW + B_IPW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;   // waves w and w + 4 sit on the same SIMD and work out of step
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const int Cin = (C::STATIC && C::CIN > 0) ? C::CIN : p.Cin;   // (CIN 0: static filter geometry, runtime channels)
    const int KH = C::STATIC ? C::KH : p.KH;
    const int KW = C::STATIC ? C::KW : p.KW;
    const int SH = C::STATIC ? C::SH : p.SH;
    const int SW = C::STATIC ? C::SW : p.SW;
    const int PH = C::STATIC ? C::PH : p.PH;
    const int PW = C::STATIC ? C::PW : p.PW;
    const int CIN8 = Cin >> 3;
    const int H = p.H, W = p.W;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int ntile = bid % p.n_ntiles;
    const int mtile = bid / p.n_ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // gather state: wave-instruction q = j * 8 + wave fills rows [16q, 16q + 16) of a slab (see conv_igemm_kernel)
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);
    const long long zero_off = zero - p.x, zero_off_w = zero - p.w;
    // BUF: static slab-aligned geometries issue their direct-to-LDS loads through buffer descriptors: the per-lane
    // offset is a 32-bit constant of the tile, the slab's position (tap, channel block / k-slab) a scalar offset, and an
    // out-of-image or tail lane is sent out of range (the load returns zeros) - two or three vector instructions per
    // load instead of the ~15 of the 64-bit address arithmetic, which made the load interval the longer half of the
    // two-group schedule.  (Needs x and w below 2 GB: ConvArgs::x_bytes, checked by the launcher.)
#ifndef SC2_CONV_NO_BUF
    constexpr bool BUF = C::STATIC && (C::CIN % 32 == 0) && C::KH * C::KW <= 32;
#else
    constexpr bool BUF = false;   // A/B build (tools/build_variant.sh nobuf -DSC2_CONV_NO_BUF)
#endif
    constexpr uint32_t OOB = 0x80000000u;
    [[maybe_unused]] buf_rsrc_t rs_x, rs_w;
    [[maybe_unused]] uint32_t a_vo[A_IPW], a_tapmask[A_IPW], b_vo[B_IPW], pw_vo[(C::PATCH_ROWS / 16 + 7) / 8];
    if constexpr (BUF) {
        const long long shift = ((long long)PH * W + PW) * Cin;   // elements: the descriptor starts at tap (0, 0) of pixel (0, 0)
        rs_x = make_rsrc(p.x - (C::PATCH3 ? 0 : shift), p.x_bytes + (C::PATCH3 ? 0u : (uint32_t)(shift * 2)));
        rs_w = make_rsrc(p.w, p.w_bytes);
        if constexpr (!C::PATCH3) {
#pragma unroll
            for (int j = 0; j < A_IPW; ++j) {
                const int m = m0 + (j * 8 + wave) * 16 + (lane >> 2);
                const bool ok = m < p.M;
                const int mm = ok ? m : 0;
                const int img = mm / p.OHW;
                const int rem = mm - img * p.OHW;
                const int oh = rem / p.OW;
                const int ow = rem - oh * p.OW;
                a_vo[j] = (uint32_t)((((long long)img * H + oh * SH) * W + ow * SW) * Cin * 2 + kc * 16);
                uint32_t mk = 0;
#pragma unroll
                for (int t = 0; t < KH * KW; ++t) {
                    const int ih = oh * SH - PH + t / KW, iw = ow * SW - PW + t % KW;
                    mk |= (ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W)) ? (1u << t) : 0u;
                }
                a_tapmask[j] = mk;
            }
        }
#pragma unroll
        for (int j = 0; j < B_IPW; ++j)
            b_vo[j] = (uint32_t)(((long long)(n0 + (j * 8 + wave) * 16 + (lane >> 2)) * p.b_row_stride + kc * 8) * 2);
    }
    long long a_off[A_IPW];
    int a_ih0[A_IPW], a_iw0[A_IPW];
    bool a_ok[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; ++j) {
        const int m = m0 + (j * 8 + wave) * 16 + (lane >> 2);
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int img = mm / p.OHW;
        const int rem = mm - img * p.OHW;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_ih0[j] = oh * SH - PH;
        a_iw0[j] = ow * SW - PW;
        a_off[j] = ((long long)(img * H + a_ih0[j]) * W + a_iw0[j]) * Cin;
    }
    long long b_off[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j)
        b_off[j] = (long long)(n0 + (j * 8 + wave) * 16 + (lane >> 2)) * p.b_row_stride + kc * 8;
    // k position of slab t.  When Cin % 32 == 0 a slab never straddles a filter tap, so the tap (kh, kw) and the
    // channel base are WAVE-UNIFORM functions of t: they live in scalar registers and cost no vector ALU; only
    // the bounds test and the final add are per lane.  Otherwise a per-lane (kh, kw, c8) state machine is stepped.
    const bool aligned = C::STATIC ? (C::CIN % 32 == 0) : (Cin % 32 == 0);
    const int spt = aligned ? (CIN8 >> 2) : 1;     // slabs per tap
    constexpr int WRAPS = C::STATIC ? (KC + (C::CIN / 8) - 1) / (C::CIN / 8 > 0 ? C::CIN / 8 : 1) : KC;
    int c8 = kc, kh = 0, kw = 0;
    auto wrap_k = [&]() {
#pragma unroll
        for (int rep = 0; rep < WRAPS; ++rep) {
            const bool w1 = c8 >= CIN8;
            c8 -= w1 ? CIN8 : 0;
            kw += w1 ? 1 : 0;
            const bool w2 = kw == KW;
            kw = w2 ? 0 : kw;
            kh += w2 ? 1 : 0;
        }
    };
    if (!aligned) wrap_k();
    const int KT = (p.dbg & 2) ? 0 : p.KT;
    int next_a = 0;   // slab index the next issue_a() call fetches

    auto issue_a = [&](int buf) {   // A rows of the next unissued slab
        unsigned char *Ab = smem + buf * C::STAGE_BYTES;
        int t_kh, t_kw;
        long long tap_off;
        if (aligned) {
            int tap, cb;   // scalar
            if (p.k_slab_major) { cb = next_a / (KH * KW); tap = next_a - cb * (KH * KW); }
            else { tap = next_a / spt; cb = next_a - tap * spt; }
            t_kh = tap / KW;
            t_kw = tap - t_kh * KW;
            tap_off = ((long long)t_kh * W + t_kw) * Cin + cb * 32 + kc * 8;
        } else {
            t_kh = kh;
            t_kw = kw;
            tap_off = ((long long)kh * W + kw) * Cin + c8 * 8;
        }
        const bool tap_ok = aligned ? (next_a < KT) : (t_kh < KH);   // false for the K tail and the dummy slabs past KT
        if constexpr (BUF) {
            int tap, cb;   // scalar: (tap, channel block) of the slab
            if (p.k_slab_major) { cb = next_a / (KH * KW); tap = next_a - cb * (KH * KW); }
            else { tap = next_a / spt; cb = next_a - tap * spt; }
            const uint32_t soff = (uint32_t)(((tap / KW) * W + tap % KW) * Cin + cb * 32) * 2u;
#pragma unroll
            for (int j = 0; j < A_IPW; ++j) {
                const uint32_t vo = (tap_ok && ((a_tapmask[j] >> tap) & 1u)) ? a_vo[j] : OOB;
                buf_load_lds16(rs_x, (lds_ptr_t)(Ab + (j * 8 + wave) * 1024), vo, soff);
            }
            ++next_a;
            return;
        }
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int ih = a_ih0[j] + t_kh, iw = a_iw0[j] + t_kw;
            const bool ok = a_ok[j] & tap_ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            const long long off = ok ? a_off[j] + tap_off : zero_off;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.x + off), (lds_ptr_t)(Ab + (j * 8 + wave) * 1024), 16, 0, 0);
        }
        ++next_a;
        if (!aligned) {
            c8 += KC;
            wrap_k();
        }
    };
    // ---- PATCH3: per-lane state of the shifted-window reads and the window fill
    [[maybe_unused]] int pr_row[MT];        // window row of (m-tile i, this lane's fragment row) at the centre tap
    [[maybe_unused]] uint32_t pr_mask[MT];  // bit (kh * 3 + kw): that tap's input pixel lies inside the image
    [[maybe_unused]] const int frow_p = lane & 15, fq_p = lane >> 4;
    [[maybe_unused]] long long g_base = 0;   // flattened NHW index of window row 0
    if constexpr (C::PATCH3) {
        if (tid < 16) reinterpret_cast<uint32_t *>(smem + C::PATCH_ZERO)[tid] = 0u;
        const int HW = H * W;
        const int m_last = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
        {   // the smallest g of the tile: its first pixel, or the first pixel of a later image of the tile
            const int img0 = m0 / p.OHW, rem0 = m0 - img0 * p.OHW;
            const int oh0 = rem0 / p.OW, ow0 = rem0 - oh0 * p.OW;
            g_base = (long long)img0 * HW + (oh0 - PH) * W + (ow0 - PW);
            const int img1 = m_last / p.OHW;
            for (int im = img0 + 1; im <= img1; ++im) {
                const long long gi = (long long)im * HW - PH * W - PW;
                g_base = gi < g_base ? gi : g_base;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wm * C::WM + i * 16 + frow_p;
            const int m = m0 + row < p.M ? m0 + row : m_last;   // (tail rows: any valid pixel, results discarded)
            const int img = m / p.OHW, rem = m - img * p.OHW;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            uint32_t mk = 0;
#pragma unroll
            for (int t = 0; t < KH * KW; ++t) {
                const int ih = oh - PH + t / KW, iw = ow - PW + t % KW;
                mk |= (((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W)) ? (1u << t) : 0u;
            }
            pr_mask[i] = mk;
            pr_row[i] = (int)((long long)img * HW + (oh - PH) * W + (ow - PW) - g_base);
        }
    }
    if constexpr (BUF && C::PATCH3) {
#pragma unroll
        for (int j = 0; j < (C::PATCH_ROWS / 16 + 7) / 8; ++j) {
            const long long g = g_base + (j * 8 + wave) * 16 + (lane >> 2);
            pw_vo[j] = ((g >= 0) & (g < (long long)p.N * H * W)) ? (uint32_t)(g * Cin * 2 + kc * 16) : OOB;
        }
    }
    auto issue_patch = [&](int cb) {   // window of channel slab cb -> patch buffer cb & 1 (this wave's rows)
        unsigned char *Pb = smem + (cb & 1) * C::PATCH_BYTES;
        const bool cb_ok = cb * 32 < Cin;
        if constexpr (BUF) {
#pragma unroll
            for (int j = 0; j < (C::PATCH_ROWS / 16 + 7) / 8; ++j) {
                const int q = j * 8 + wave;
                if (q < C::PATCH_ROWS / 16)   // wave-uniform
                    buf_load_lds16(rs_x, (lds_ptr_t)(Pb + q * 1024), cb_ok ? pw_vo[j] : OOB, (uint32_t)cb * 64u);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < (C::PATCH_ROWS / 16 + 7) / 8; ++j) {
            const int q = j * 8 + wave;
            if (q < C::PATCH_ROWS / 16) {   // wave-uniform
                const long long g = g_base + q * 16 + (lane >> 2);   // flattened NHW input pixel
                const bool ok = cb_ok & (g >= 0) & (g < (long long)p.N * H * W);
                const long long off = ok ? g * Cin + cb * 32 + kc * 8 : zero_off;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.x + off), (lds_ptr_t)(Pb + q * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_b = [&](int kt, int buf) {
        unsigned char *Bb = C::PATCH3 ? smem + C::PATCH_B0 + buf * C::B_BYTES : smem + buf * C::STAGE_BYTES + C::A_BYTES;
        const bool kt_ok = kt < KT;
        if constexpr (BUF) {
            const uint32_t soff = (uint32_t)(kt_ok ? kt : KT - 1) * (uint32_t)p.b_kt_stride * 2u;   // (slabs past KT are never read)
#pragma unroll
            for (int j = 0; j < B_IPW; ++j)
                buf_load_lds16(rs_w, (lds_ptr_t)(Bb + (j * 8 + wave) * 1024), b_vo[j], soff);
            return;
        }
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const long long off = kt_ok ? b_off[j] + (long long)kt * p.b_kt_stride : zero_off_w;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.w + off), (lds_ptr_t)(Bb + (j * 8 + wave) * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const uint32_t amask = p.aop == SC2_AOP_ABS ? 0x7FFF7FFFu : 0xFFFFFFFFu;
    const int frow = lane & 15, fq = lane >> 4;
    uint32_t a_rd[MT], b_rd[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rd[i] = (uint32_t)lds_off(wm * C::WM + i * 16 + frow, fq);
#pragma unroll
    for (int j = 0; j < NT; ++j)
        b_rd[j] = (uint32_t)((C::PATCH3 ? C::PATCH_B0 : C::A_BYTES) + lds_off(wn * C::WN + j * 16 + frow, fq));

    if constexpr (C::PATCH3) issue_patch(0);
#pragma unroll
    for (int st = 0; st < S - 1; ++st) {
        if constexpr (!C::PATCH3) issue_a(st);
        issue_b(st, st);
    }
    constexpr int LW = C::PATCH3 ? B_IPW : L;   // counted loads per slab (the window loads are older than any slab waited for)
    [[maybe_unused]] int p_cb = 0, p_tap = 0;   // PATCH3: channel slab and tap of k-step kt
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LW) : "memory");   // slab 0 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

    uint4 bv[NT];
    for (int kt = 0; kt < KT; ++kt) {
        const uint32_t sb = lds_base + (uint32_t)((kt % S) * (C::PATCH3 ? C::B_BYTES : C::STAGE_BYTES));
        const int nbuf = (kt + S - 1) % S;
        [[maybe_unused]] const uint32_t pbase = lds_base + (uint32_t)((p_cb & 1) * C::PATCH_BYTES);
        [[maybe_unused]] const int p_shift = (p_tap / KW) * W + p_tap % KW;
#pragma unroll
        for (int ph = 0; ph < PHASES; ++ph) {
            // ---- load interval: this phase's fragments, a share of slab kt+S-1's loads
            uint4 av[4];
            if (ph == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = lds_read16(sb + b_rd[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (C::PATCH3) {
                    const int r = pr_row[4 * ph + i] + p_shift;
                    const uint32_t in_img = (pr_mask[4 * ph + i] >> p_tap) & 1u;
                    const uint32_t addr = in_img ? pbase + (uint32_t)lds_off(r, fq_p) : lds_base + C::PATCH_ZERO + fq_p * 16;
                    av[i] = lds_read16(addr);
                } else {
                    av[i] = lds_read16(sb + a_rd[4 * ph + i]);
                }
            }
            if constexpr (C::PATCH3) {
                if (ph == 0 && p_tap == 0) issue_patch(p_cb + 1);   // next slab's window (its buffer was last read a slab ago)
            } else {
                if (ph == 0) issue_a(nbuf);
            }
            if (ph == PHASES - 1) issue_b(kt + S - 1, nbuf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ph == PHASES - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LW) : "memory");   // slab kt+1 landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- MFMA interval (the partner group is in its load interval)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            bf16x8_t af[4], bfr[NT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = av[i];
                if (p.aop == SC2_AOP_ABS) {   // uniform: only the GDN GEMMs take |x|
                    v.x &= amask; v.y &= amask; v.z &= amask; v.w &= amask;
                }
                af[i] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, bv[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[4 * ph + i][j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[4 * ph + i][j], 0, 0, 0);   // D = W X^T
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (C::PATCH3) {
            if (++p_tap == KH * KW) { p_tap = 0; ++p_cb; }
        }
    }
    if (group == 0) __builtin_amdgcn_s_barrier();   // re-align the groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // dummy slabs past KT
    __builtin_amdgcn_s_barrier();

    if (p.dbg & 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    conv_big_epilogue<C, 512>(p, smem, acc, tid, lane, wm, wn, frow, fq, m0, n0);
}

// ======================================================================================================
// Register-tile variant of the big tile: 256 threads = 4 waves, ONE per SIMD with the whole 512-entry register file,
// each owning a 128 x 128 quarter of the 256 x 256 tile (64 accumulator tiles = 256 registers).  Per 32-deep slab a
// wave issues 64 MFMAs against 16 fragment reads (8 pixel + 8 weight fragments): a quarter of the LDS bytes per MFMA
// of the 8-wave tiling (128 x 64 per wave: 12 reads per 32 MFMAs), which is what bounds that kernel - its reads plus
// the direct-to-LDS writes need more LDS cycles per slab than its MFMAs need matrix-pipe cycles.  No partner wave
// hides latency here, so the slab loop is software-pipelined inside the wave: the fragments of slab t + 1 are read
// into a second register set BETWEEN the MFMAs of slab t (one read per four MFMAs), the direct-to-LDS loads of slab
// t + 3 are issued at the top of slab t, and there is ONE barrier per slab (1024 matrix-pipe cycles).
//   RAW: every wave retires its share of slab t + 1 (counted vmcnt) before the barrier at the top of slab t, and
//        reads that slab only behind it.
//   WAR: slab t + 3 lands in the stage slab t - 1 lived in; its fragments were read during slab t - 2 and waited
//        for (lgkmcnt(0)) before the barrier at the top of slab t - 1.
template <bool STATIC_, int CIN_, int KH_, int KW_, int SH_, int SW_, int PH_, int PW_>
struct Cfg4 {
    static constexpr bool SQ = false;
    static constexpr int BM = 256, BN = 256, BK = 32, KC = 4;
    static constexpr int WAVES_M = 2, WAVES_N = 2;
    static constexpr bool STATIC = STATIC_;
    static constexpr int CIN = CIN_, KH = KH_, KW = KW_, SH = SH_, SW = SW_, PH = PH_, PW = PW_;
    static constexpr int WM = 128, WN = 128, MT = 8, NT = 8;
    static constexpr int STAGES = 4;
    static constexpr int A_IPW = BM / 16 / 4, B_IPW = BN / 16 / 4;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int STAGE_ROWS = WAVES_M * 16;
    static constexpr int MAIN_LDS = STAGES * STAGE_BYTES;
    static constexpr int EPI_LDS = STAGE_ROWS * (BN + 4) * 4;
    static constexpr int FUSE_LDS = BM * 512 + 32768;
    static constexpr int IMG_LDS = BM * BN * 2;
    static constexpr int LDS_BYTES = FUSE_LDS;   // the largest of the four
    static_assert(FUSE_LDS >= MAIN_LDS && FUSE_LDS >= EPI_LDS && FUSE_LDS >= IMG_LDS, "LDS plan");
};

template <class C>
__global__ __launch_bounds__(256, 1) void conv_igemm4_kernel(const ConvArgs p) {
    constexpr int BM = C::BM, BN = C::BN, KC = C::KC;
    constexpr int MT = C::MT, NT = C::NT, S = C::STAGES;
    constexpr int A_IPW = C::A_IPW, B_IPW = C::B_IPW, L = A_IPW + B_IPW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int Cin = C::STATIC ? C::CIN : p.Cin;
    const int KH = C::STATIC ? C::KH : p.KH;
    const int KW = C::STATIC ? C::KW : p.KW;
    const int SH = C::STATIC ? C::SH : p.SH;
    const int SW = C::STATIC ? C::SW : p.SW;
    const int PH = C::STATIC ? C::PH : p.PH;
    const int PW = C::STATIC ? C::PW : p.PW;
    const int CIN8 = Cin >> 3;
    const int H = p.H, W = p.W;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int ntile = bid % p.n_ntiles;
    const int mtile = bid / p.n_ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // gather state: wave-instruction q = j * 4 + wave fills rows [16q, 16q + 16) of a slab (see conv_igemm_kernel)
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);
    const long long zero_off = zero - p.x, zero_off_w = zero - p.w;
    long long a_off[A_IPW];
    int a_ih0[A_IPW], a_iw0[A_IPW];
    bool a_ok[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; ++j) {
        const int m = m0 + (j * 4 + wave) * 16 + (lane >> 2);
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int img = mm / p.OHW;
        const int rem = mm - img * p.OHW;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_ih0[j] = oh * SH - PH;
        a_iw0[j] = ow * SW - PW;
        a_off[j] = ((long long)(img * H + a_ih0[j]) * W + a_iw0[j]) * Cin;
    }
    long long b_off[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j)
        b_off[j] = (long long)(n0 + (j * 4 + wave) * 16 + (lane >> 2)) * p.b_row_stride + kc * 8;
    const bool aligned = C::STATIC ? (C::CIN % 32 == 0) : (Cin % 32 == 0);
    const int spt = aligned ? (CIN8 >> 2) : 1;     // slabs per tap
    constexpr int WRAPS = C::STATIC ? (KC + (C::CIN / 8) - 1) / (C::CIN / 8 > 0 ? C::CIN / 8 : 1) : KC;
    int c8 = kc, kh = 0, kw = 0;
    auto wrap_k = [&]() {
#pragma unroll
        for (int rep = 0; rep < WRAPS; ++rep) {
            const bool w1 = c8 >= CIN8;
            c8 -= w1 ? CIN8 : 0;
            kw += w1 ? 1 : 0;
            const bool w2 = kw == KW;
            kw = w2 ? 0 : kw;
            kh += w2 ? 1 : 0;
        }
    };
    if (!aligned) wrap_k();
    const int KT = (p.dbg & 2) ? 0 : p.KT;
    int next_a = 0;   // slab index the next issue_a() call fetches

    auto issue_a = [&](int buf) {   // A rows of the next unissued slab
        unsigned char *Ab = smem + buf * C::STAGE_BYTES;
        int t_kh, t_kw;
        long long tap_off;
        if (aligned) {
            int tap, cb;   // scalar
            if (p.k_slab_major) { cb = next_a / (KH * KW); tap = next_a - cb * (KH * KW); }
            else { tap = next_a / spt; cb = next_a - tap * spt; }
            t_kh = tap / KW;
            t_kw = tap - t_kh * KW;
            tap_off = ((long long)t_kh * W + t_kw) * Cin + cb * 32 + kc * 8;
        } else {
            t_kh = kh;
            t_kw = kw;
            tap_off = ((long long)kh * W + kw) * Cin + c8 * 8;
        }
        const bool tap_ok = aligned ? (next_a < KT) : (t_kh < KH);   // false for the K tail and the dummy slabs past KT
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int ih = a_ih0[j] + t_kh, iw = a_iw0[j] + t_kw;
            const bool ok = a_ok[j] & tap_ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            const long long off = ok ? a_off[j] + tap_off : zero_off;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.x + off), (lds_ptr_t)(Ab + (j * 4 + wave) * 1024), 16, 0, 0);
        }
        ++next_a;
        if (!aligned) {
            c8 += KC;
            wrap_k();
        }
    };
    auto issue_b = [&](int kt, int buf) {
        unsigned char *Bb = smem + buf * C::STAGE_BYTES + C::A_BYTES;
        const bool kt_ok = kt < KT;
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const long long off = kt_ok ? b_off[j] + (long long)kt * p.b_kt_stride : zero_off_w;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.w + off), (lds_ptr_t)(Bb + (j * 4 + wave) * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const uint32_t amask = p.aop == SC2_AOP_ABS ? 0x7FFF7FFFu : 0xFFFFFFFFu;
    const int frow = lane & 15, fq = lane >> 4;
    // fragment (i, lane) of the wave's rows sits 1024 B after fragment (i - 1, lane): one base + immediates
    const uint32_t a_rd = lds_base + (uint32_t)lds_off(wm * C::WM + frow, fq);
    const uint32_t b_rd = lds_base + (uint32_t)(C::A_BYTES + lds_off(wn * C::WN + frow, fq));
    auto read_frag = [](uint32_t addr, int imm) {   // ds_read_b128 with the tile offset as an immediate
        uint4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0) : "memory");
        (void)imm;
        return v;
    };
    (void)read_frag;

#pragma unroll
    for (int st = 0; st < S - 1; ++st) {
        issue_a(st);
        issue_b(st, st);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");   // slab 0 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();

    uint4 fa[2][MT], fb[2][NT];   // fragment register sets: slab t in set t & 1
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[0][i] = lds_read16(a_rd + i * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[0][j] = lds_read16(b_rd + j * 1024);

    auto slab = [&](int kt, uint4 (&ca)[MT], uint4 (&cb)[NT], uint4 (&na)[MT], uint4 (&nb)[NT]) {
        // slab kt + 1 landed (this wave's share), this wave's reads of slab kt are back: then everybody's are
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 3) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nbuf = (kt + S - 1) % S;
        issue_a(nbuf);
        issue_b(kt + S - 1, nbuf);
        const uint32_t so = (uint32_t)(((kt + 1) % S) * C::STAGE_BYTES);
        bf16x8_t bw[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bw[j] = __builtin_bit_cast(bf16x8_t, cb[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            uint4 v = ca[i];
            v.x &= amask; v.y &= amask; v.z &= amask; v.w &= amask;
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
            // two fragment reads of the next slab per row of eight MFMAs
            na[i] = lds_read16(a_rd + so + i * 1024);
#pragma unroll
            for (int j = 0; j < NT / 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], af, acc[i][j], 0, 0, 0);   // D = W X^T
            __builtin_amdgcn_sched_barrier(0);
            nb[i] = lds_read16(b_rd + so + i * 1024);
#pragma unroll
            for (int j = NT / 2; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], af, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(MT == NT, "one B read per A row");
    for (int kt = 0; kt < KT; kt += 2) {
        slab(kt, fa[0], fb[0], fa[1], fb[1]);
        if (kt + 1 < KT) slab(kt + 1, fa[1], fb[1], fa[0], fb[0]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // dummy slabs past KT, the last prefetched fragments
    __builtin_amdgcn_s_barrier();

    if (p.dbg & 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    conv_big_epilogue<C, 256>(p, smem, acc, tid, lane, wm, wn, frow, fq, m0, n0);
}

template <class C>
int launch4(const ConvArgs &a, hipStream_t s) {
    ConvArgs p = a;
    p.KT = (a.KH * a.KW * a.Cin + C::BK - 1) / C::BK;
    p.n_ntiles = (a.Cout + C::BN - 1) / C::BN;
    const int n_mtiles = (a.M + C::BM - 1) / C::BM;
    const long long nwg = (long long)n_mtiles * p.n_ntiles;
    if (nwg <= 0 || nwg > 0x7FFFFFFFLL) {
        sc2_set_error("conv2d: grid of %lld workgroups out of range", nwg);
        return SC2_ERR_INVALID_ARG;
    }
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm4_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_igemm4_kernel<C>, dim3((unsigned)nwg), dim3(256), C::LDS_BYTES, s, p);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

template <class C>
int launch8(const ConvArgs &a, hipStream_t s) {
    ConvArgs p = a;
    p.KT = (a.KH * a.KW * a.Cin + C::BK - 1) / C::BK;
    p.n_ntiles = (a.Cout + C::BN - 1) / C::BN;
    const int n_mtiles = (a.M + C::BM - 1) / C::BM;
    const long long nwg = (long long)n_mtiles * p.n_ntiles;
    if (nwg <= 0 || nwg > 0x7FFFFFFFLL) {
        sc2_set_error("conv2d: grid of %lld workgroups out of range", nwg);
        return SC2_ERR_INVALID_ARG;
    }
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm8_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_igemm8_kernel<C>, dim3((unsigned)nwg), dim3(512), C::LDS_BYTES, s, p);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

template <class C>
int launch(const ConvArgs &a, hipStream_t s) {
    ConvArgs p = a;
    p.KT = (a.KH * a.KW * a.Cin + C::BK - 1) / C::BK;
    p.n_ntiles = (a.Cout + C::BN - 1) / C::BN;
    const int n_mtiles = (a.M + C::BM - 1) / C::BM;
    const long long nwg = (long long)n_mtiles * p.n_ntiles;
    if (nwg <= 0 || nwg > 0x7FFFFFFFLL) {
        sc2_set_error("conv2d: grid of %lld workgroups out of range", nwg);
        return SC2_ERR_INVALID_ARG;
    }
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_igemm_kernel<C>, dim3((unsigned)nwg), dim3(256), C::LDS_BYTES, s, p);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

template <class C>
int launch_patch(const ConvArgs &a, hipStream_t s) {
    ConvArgs p = a;
    p.KT = 0;
    p.n_ntiles = 1;
    const int J = a.OW + 2;
    const int patch = 2 * 7 * J * 64;
    const int lds = patch > C::LDS_BYTES ? patch : C::LDS_BYTES;
    const long long nwg = (long long)a.N * ((a.OH + 1) / 2);
    if (nwg <= 0 || nwg > 0x7FFFFFFFLL) {
        sc2_set_error("conv2d: grid of %lld workgroups out of range", nwg);
        return SC2_ERR_INVALID_ARG;
    }
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv5s2_patch_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 7 * 66 * 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv5s2_patch_kernel<C>, dim3((unsigned)nwg), dim3(256), lds, s, p);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

// Static geometries of FPBasedResNetBottleneck(24, 256) (layer.py:464-494) -> folded address math.
//                 BM   BN  WM WN  static Cin KH KW SH SW PH PW
using C_conv0 = Cfg<128, 96, 2, 2, true, 8, 5, 3, 2, 1, 2, 1>;      // 3->96 k5 s2 p2 on the pixel-pair view
using C_gdn96 = Cfg<128, 96, 2, 2, true, 96, 1, 1, 1, 1, 0, 0>;
using C_conv2 = Cfg<128, 48, 4, 1, true, 96, 5, 5, 2, 2, 2, 2, 3>;     // 96->48 k5 s2 p2
using C_gdn48 = Cfg<128, 48, 4, 1, true, 48, 1, 1, 1, 1, 0, 0>;
using C_conv4 = Cfg<128, 32, 4, 1, true, 48, 2, 2, 1, 1, 0, 0>;     // 48->24 k2
using C_dec0 = Cfg<128, 128, 2, 2, true, 24, 2, 2, 1, 1, 1, 1>;     // 24->512 k2 p1
using C_gdn512 = Cfg<128, 128, 2, 2, true, 512, 1, 1, 1, 1, 0, 0, 3>;
using C_dec2 = Cfg<128, 128, 2, 2, true, 512, 2, 2, 1, 1, 0, 0, 3>;    // 512->256 k2
using C_gdn256 = Cfg<128, 128, 2, 2, true, 256, 1, 1, 1, 1, 0, 0, 3>;
using C_dec4 = Cfg<128, 128, 2, 2, true, 256, 2, 2, 1, 1, 1, 1, 3>;    // 256->256 k2 p1
// Runtime-geometry fallbacks (other channel widths, the ResNet tail, other bottleneck sizes).
using G_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3>;
using G_96 = Cfg<128, 96, 2, 2, false, 0, 0, 0, 0, 0, 0, 0>;
using G_64 = Cfg<128, 64, 2, 2, false, 0, 0, 0, 0, 0, 0, 0>;
using G_48 = Cfg<128, 48, 4, 1, false, 0, 0, 0, 0, 0, 0, 0>;
using G_32 = Cfg<128, 32, 4, 1, false, 0, 0, 0, 0, 0, 0, 0>;

// twins that prefetch the epilogue operand (GDN / IGDN on x, residual add)
using Cx_gdn96 = Cfg<128, 96, 2, 2, true, 96, 1, 1, 1, 1, 0, 0, 2, true>;
using Cx_gdn48 = Cfg<128, 48, 4, 1, true, 48, 1, 1, 1, 1, 0, 0, 2, true>;
using Cx_gdn512 = Cfg<128, 128, 2, 2, true, 512, 1, 1, 1, 1, 0, 0, 3, true>;
using Cx_gdn256 = Cfg<128, 128, 2, 2, true, 256, 1, 1, 1, 1, 0, 0, 3, true>;
using Gx_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3, true>;
using Gx_96 = Cfg<128, 96, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 2, true>;
// squared-form GDN GEMMs (1x1 on x^2, rsqrt / sqrt epilogue): bmshj2018_factorized, N = 128 / 192 (rows 128 / 256)
using Gq_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3, false, true>;
// dilated (atrous) convolutions: the generic 128-wide tile with the runtime dilation compiled in (DeepLab's ASPP branches,
// sc2bench/models/segmentation/deeplabv3.py: rates 12 / 24 / 36 on the 2048-channel map; torchvision's dilated layer3 / layer4)
using Gd_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3, false, false, true>;
// the two GEMMs of the GDN1 backward with its element-wise halves in their epilogues (sc2_gdn1_bwd_gemm): C = 256 / 512, C = 96
using Gb_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3, false, false, false, true>;
using Gb_96 = Cfg<128, 96, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 2, false, false, false, true>;
using Gqx_128 = Cfg<128, 128, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 3, true, true>;
using Gx_64 = Cfg<128, 64, 2, 2, false, 0, 0, 0, 0, 0, 0, 0, 2, true>;
using Gx_48 = Cfg<128, 48, 4, 1, false, 0, 0, 0, 0, 0, 0, 0, 2, true>;
using Gx_32 = Cfg<128, 32, 4, 1, false, 0, 0, 0, 0, 0, 0, 0, 2, true>;

// big-tile (8-wave) geometries: the MFMA-bound decoder layers and the runtime-geometry fallback
using B_gdn512 = Cfg8<256, 2, 4, true, 512, 1, 1, 1, 1, 0, 0>;
using B_dec2 = Cfg8<256, 2, 4, true, 512, 2, 2, 1, 1, 0, 0>;
using B_gdn256 = Cfg8<256, 2, 4, true, 256, 1, 1, 1, 1, 0, 0>;
using B_dec4 = Cfg8<256, 2, 4, true, 256, 2, 2, 1, 1, 1, 1>;
using H_dec2 = Cfg8<256, 2, 4, true, 512, 2, 2, 1, 1, 0, 0, 128, 3>;   // half-height twins: 2 workgroups per CU
using H_dec4 = Cfg8<256, 2, 4, true, 256, 2, 2, 1, 1, 1, 1, 128, 3>;
using BG_256 = Cfg8<256, 2, 4, false, 0, 0, 0, 0, 0, 0, 0>;
// register-tile (4-wave, 128 x 128 per wave) geometries
using R_dec2 = Cfg4<true, 512, 2, 2, 1, 1, 0, 0>;
using R_dec4 = Cfg4<true, 256, 2, 2, 1, 1, 1, 1>;
using RG_256 = Cfg4<false, 0, 0, 0, 0, 0, 0, 0>;
using BG_128 = Cfg8<128, 4, 2, false, 0, 0, 0, 0, 0, 0, 0>;
// 3x3 stride-1 pad-1 layers of the ResNet tail: the nine taps read one staged window (Cfg8::PATCH3)
using P3_256 = Cfg8<256, 2, 4, true, 0, 3, 3, 1, 1, 1, 1, 256, 4, true>;
using P3_128 = Cfg8<128, 4, 2, true, 0, 3, 3, 1, 1, 1, 1, 256, 4, true>;
// 3x3 stride-2 pad-1 layers (first block of each stage of the tail): im2col gather with buffer-addressed loads
using S2_128 = Cfg8<128, 4, 2, true, 0, 3, 3, 2, 2, 1, 1>;
using S2_256 = Cfg8<256, 2, 4, true, 0, 3, 3, 2, 2, 1, 1>;
// 2x2 stride-1 decoder layers (dec.conv2: pad 0, dec.conv4: pad 1): windows of up to 448 pixels (tiles that cross an image)
using P2_dec2 = Cfg8<256, 2, 4, true, 0, 2, 2, 1, 1, 0, 0, 256, 4, true, 192>;
using P2_dec4 = Cfg8<256, 2, 4, true, 0, 2, 2, 1, 1, 1, 1, 256, 4, true, 192>;

template <class C>
bool matches(const ConvArgs &a) {
    return a.Cin == C::CIN && a.KH == C::KH && a.KW == C::KW && a.SH == C::SH && a.SW == C::SW && a.PH == C::PH &&
           a.PW == C::PW;
}


}  // namespace sc2conv
