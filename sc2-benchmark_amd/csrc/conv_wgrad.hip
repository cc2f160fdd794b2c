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
// through a 4-deep LDS ring (counted vmcnt, one barrier per slab; a slab's fragments are read one slab ahead of its MFMAs).
// Round 5, CT = 256: layers with more than 128 output channels take a 256-channel x 128-column tile (8 waves, one workgroup per CU).
// The 128 x 128 tile moves 16 KB from L2 into LDS per 64 MFMAs = 64 flop / B: at the ~17 - 19 TB/s the eight L2s deliver into LDS
// (MI355X_MICROARCH.md, "Indexed rows: gather into LDS") that alone caps the launch at 1.1 - 1.2 PFLOP/s, beside the matrix pipe's own
// limit (measured 0.78 - 0.87).  256 x 128 moves 24 KB per 128 MFMAs = 85 flop / B.  The pixel range
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

constexpr int WG_TILE = 128;              // k-columns per tile (and channels per tile of the CT = 128 form)
constexpr int WG_ROWB = WG_TILE * 2;      // bytes per pixel row of the im2col slab image
constexpr int WG_SLAB = 32;               // pixels per slab
constexpr int WG_IMG = WG_SLAB * WG_ROWB; // 8 KB: the im2col image of a slab
constexpr int WG_STAGES = 4;            // (round 5: four, since a slab's fragments are read one slab ahead of its MFMAs)
__device__ __forceinline__ int wg_swz(int row) { return 2 * ((row & 3) + 4 * ((row >> 3) & 1)); }
// ... of a 128-byte row (the 64-channel dY image, 8 chunks): two rows per 256 bytes of banks, so row parity already separates two of
// the eight rows a fragment read touches; the XOR spreads the other four: 32-byte slot = (4 row + ((chunk ^ f) >> 1)) mod 8
__device__ __forceinline__ int wg_swz128(int row) { return 2 * (((row >> 1) & 1) + 2 * ((row >> 3) & 1)); }
template <int CT>
struct WgGeo {                            // CT = output channels per tile: 64 (2 waves, three workgroups per CU), 128 (4 waves, two) or 256 (8 waves, one)
    static constexpr int NW = CT / 32;                    // waves: (CT / 64) x 2, 64 x 64 each
    static constexpr int G_ROWB = CT * 2;                 // bytes per pixel row of the dY slab image
    static constexpr int G_IMG = WG_SLAB * G_ROWB;        // 8 / 16 KB
    static constexpr int STAGE = WG_IMG + G_IMG;          // a ring stage: [im2col image][dY image]
    static constexpr int G_RPI = 1024 / G_ROWB;           // dY rows per 1 KB direct-to-LDS instruction (4 / 2)
    static constexpr int G_CPR = G_ROWB / 16;             // 16-byte chunks per dY row (16 / 32)
    static constexpr int GJ = G_IMG / (NW * 1024);        // dY instructions per wave and slab (2)
    static constexpr int AJ = WG_IMG / (NW * 1024);       // im2col instructions per wave and slab (4 / 2 / 1)
    static __device__ __forceinline__ int gswz(int row) { return CT == 64 ? wg_swz128(row) : wg_swz(row); }
    static constexpr int L = GJ + AJ;                     // vector-memory operations per wave and slab
    static_assert(GJ == 2 && (AJ == 4 || AJ == 2 || AJ == 1) && (WG_STAGES - 1) * STAGE + WG_IMG + G_IMG <= 160 * 1024, "geometry");
    // stage bases are ds immediates (16 bits): stages whose images end above 64 KB are addressed from a second base (HI_FROM)
    static constexpr int HI_FROM = 2;
    static_assert((HI_FROM - 1) * STAGE + WG_IMG + G_IMG <= 65536 && (WG_STAGES - 1 - HI_FROM) * STAGE + WG_IMG + G_IMG <= 65536, "immediates");
};


template <int OFF>
__device__ __forceinline__ uint2 lds_read_tr(uint32_t addr) {   // (stage base as an immediate: no address arithmetic per read)
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <bool ABS, int CT>   // |x| as the im2col operand (d gamma of GDN1): a compile-time property (16 mask operations per slab otherwise)
__global__ __launch_bounds__(64 * WgGeo<CT>::NW, CT == 256 ? 1 : 2) void conv_wgrad_kernel(const WgradArgs p) {
    using G = WgGeo<CT>;
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
    const int co0 = ctile * CT, k0 = ktile * WG_TILE;
    const int m_begin = chunk * p.rows_per_block;
    const int m_end = min(p.M, m_begin + p.rows_per_block);
    const int n_slabs = (m_end - m_begin + WG_SLAB - 1) / WG_SLAB;

    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_wzero16);
    const long long zoff_x = zero - p.x, zoff_g = zero - p.gy;

    // direct-to-LDS assignment.  im2col image (256-byte rows): wave-instruction q = j * NW + wave (j < AJ) covers slab rows
    // [4q, 4q + 4); lane l -> row 4q + (l >> 4), stored chunk l & 15, which holds logical chunk (l & 15) ^ swz(row).  dY image
    // (CT * 2-byte rows): instruction q = j * NW + wave (j < 2) covers rows [G_RPI q, G_RPI (q + 1)); lane l -> row G_RPI q +
    // l / G_CPR, stored chunk l % G_CPR, logical chunk (that) ^ swz(row) (the XOR stays inside a 256-byte half of the row).
    int arow_j[G::AJ], kh_j[G::AJ], kw_j[G::AJ], ci_j[G::AJ];
    bool kok_j[G::AJ];
#pragma unroll
    for (int j = 0; j < G::AJ; ++j) {
        const int row = (j * G::NW + wave) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ wg_swz(row);
        arow_j[j] = row;
        const int k = k0 + 8 * c;
        kok_j[j] = k < p.K;
        const int kk = kok_j[j] ? k : 0;
        const int tap = kk / p.Cin;
        ci_j[j] = kk - tap * p.Cin;
        kh_j[j] = tap / p.KW;
        kw_j[j] = tap - kh_j[j] * p.KW;
    }
    int grow_j[G::GJ], gco_j[G::GJ];
    bool gok_j[G::GJ];
#pragma unroll
    for (int j = 0; j < G::GJ; ++j) {
        const int row = (j * G::NW + wave) * G::G_RPI + lane / G::G_CPR;
        const int c = (lane % G::G_CPR) ^ G::gswz(row);
        grow_j[j] = row;
        gco_j[j] = co0 + 8 * c;
        gok_j[j] = gco_j[j] < p.Cout;
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
    int m_j[G::AJ], oh_j[G::AJ], ow_j[G::AJ], ih_j[G::AJ], iw_j[G::AJ];
    uint32_t aoff_j[G::AJ];
#pragma unroll
    for (int j = 0; j < G::AJ; ++j) {
        m_j[j] = m_begin + arow_j[j];
        const int mm = m_j[j] < p.M ? m_j[j] : 0;      // (rows past the end: coordinates of pixel 0, never used)
        const int img = mm / p.OHW;
        const int rem = mm - img * p.OHW;
        oh_j[j] = rem / p.OW;
        ow_j[j] = rem - oh_j[j] * p.OW;
        ih_j[j] = oh_j[j] * p.SH - p.PH + kh_j[j];
        iw_j[j] = ow_j[j] * p.SW - p.PW + kw_j[j];
        aoff_j[j] = (uint32_t)(((img * p.H + ih_j[j]) * p.W + iw_j[j]) * p.Cin + ci_j[j]);
    }
    int mg_j[G::GJ];
    uint32_t goff_j[G::GJ];
#pragma unroll
    for (int j = 0; j < G::GJ; ++j) {
        mg_j[j] = m_begin + grow_j[j];
        goff_j[j] = (uint32_t)(mg_j[j] * p.Cout + gco_j[j]);
    }
    auto issue_slab = [&](int buf) {   // the NEXT slab in order
        unsigned char *Ai = smem + buf * G::STAGE;
        unsigned char *Gi = Ai + WG_IMG;
#pragma unroll
        for (int j = 0; j < G::GJ; ++j) {
            // dY operand
            const bool g_ok = (mg_j[j] < m_end) & gok_j[j];
            const long long goff = g_ok ? (long long)goff_j[j] : zoff_g;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.gy + goff), (lds_ptr_t)(Gi + (j * G::NW + wave) * 1024), 16, 0, 0);
            mg_j[j] += WG_SLAB;
            goff_j[j] += g_step;
        }
#pragma unroll
        for (int j = 0; j < G::AJ; ++j) {
            const bool mok = m_j[j] < m_end;
            // im2col operand
            const bool a_ok = mok & kok_j[j] & ((unsigned)ih_j[j] < (unsigned)p.H) & ((unsigned)iw_j[j] < (unsigned)p.W);
            const long long aoff = a_ok ? (long long)aoff_j[j] : zoff_x;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.x + aoff), (lds_ptr_t)(Ai + (j * G::NW + wave) * 1024), 16, 0, 0);
            // 32 pixels on
            m_j[j] += WG_SLAB;
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
            const int gc = wm * 64 + tt * 16 + 4 * (i16 & 3);   // column inside the CT-wide dY image
            const int ac = wn * 64 + tt * 16 + 4 * (i16 & 3);   // ... the 128-wide im2col image
            g_rd[tt][h] = lds_base + (uint32_t)(r * G::G_ROWB + (((gc >> 3) ^ G::gswz(r)) << 4) + ((gc >> 2) & 1) * 8);
            a_rd[tt][h] = lds_base + (uint32_t)(r * WG_ROWB + (((ac >> 3) ^ wg_swz(r)) << 4) + ((ac >> 2) & 1) * 8);
        }

    constexpr uint32_t xmask = 0x7FFF7FFFu;
    constexpr int S = WG_STAGES, L = G::L;
    static_assert(S == 4, "the slab loop below is unrolled by the ring depth (and by two fragment sets)");
    // Round 5: a slab's fragments are read from LDS ONE SLAB AHEAD, behind the barrier that publishes it and in front of the previous
    // slab's MFMAs (two fragment sets): before, a wave read its sixteen fragments, waited for them, and only then issued its sixteen
    // MFMAs -- the LDS round trip of every slab sat in front of its matrix work, hidden only by whatever the SIMD's other wave was
    // doing.  Order of a step (slab s in set P, ring stage ST):
    //     loads of slab s + 3 -> stage (ST + 3) % 4   (last read -- slab s - 1 -- before the previous step's barrier)
    //     vmcnt(2 L): slab s + 1 has landed;  lgkmcnt(0): the fragments of slab s are in their registers
    //     barrier:  every wave's part of slab s + 1 is in LDS, and every wave has finished READING slab s
    //     fragment reads of slab s + 1 -> set P ^ 1;   16 MFMAs on set P
    // (the fragment registers are written by asm reads the compiler does not track: nothing may touch a set between its reads and
    //  the lgkmcnt(0) of the next step -- tools/audit_lds_inflight.py follows them through the listing)
    uint2 gv[2][4][2], av[2][4][2];
    auto read_frags = [&](auto stage_c, auto set_c) {
        constexpr int ST = decltype(stage_c)::value, P = decltype(set_c)::value;
        constexpr int BASE = (ST >= G::HI_FROM ? ST - G::HI_FROM : ST) * G::STAGE;
        constexpr int AOFF = BASE, GOFF = BASE + WG_IMG;
        constexpr uint32_t HI = ST >= G::HI_FROM ? (uint32_t)(G::HI_FROM * G::STAGE) : 0u;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gv[P][tt][h] = lds_read_tr<GOFF>(g_rd[tt][h] + HI);
                av[P][tt][h] = lds_read_tr<AOFF>(a_rd[tt][h] + HI);
            }
    };
#pragma unroll
    for (int st = 0; st < S - 1; ++st) issue_slab(st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");   // slab 0 has landed
    __builtin_amdgcn_s_barrier();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

    // one slab out of ring stage ST, fragments in set P (compile-time: stage bases are immediates, sets are registers)
    auto slab = [&](auto stage_c, auto set_c) {
        constexpr int ST = decltype(stage_c)::value, P = decltype(set_c)::value;
        issue_slab((ST + S - 1) % S);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((S - 2) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t gf[4], af[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            gf[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(gv[P][tt][0].x, gv[P][tt][0].y, gv[P][tt][1].x, gv[P][tt][1].y));
            if (ABS) af[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(av[P][tt][0].x & xmask, av[P][tt][0].y & xmask, av[P][tt][1].x & xmask,
                                                                      av[P][tt][1].y & xmask));
            else af[tt] = __builtin_bit_cast(bf16x8_t, make_uint4(av[P][tt][0].x, av[P][tt][0].y, av[P][tt][1].x, av[P][tt][1].y));
        }
        read_frags(std::integral_constant<int, (ST + 1) % S>{}, std::integral_constant<int, P ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], af[j], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // whole trips of four slabs, no tail code (a fragment set written in one conditional block and consumed in the next would be a
    // value the compiler may copy at the join -- in flight): the launcher makes a range a multiple of 128 pixels, and slabs past the
    // end of the LAST range read the zero line
#pragma unroll 1
    for (int sl = 0; sl < n_slabs; sl += S) {
        slab(I0{}, I0{});
        slab(I1{}, I1{});
        slab(I2{}, I0{});
        slab(I3{}, I1{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragment reads of the slab past the last one)
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
    // (round 5) more than 128 output channels: the 256-channel tile, 8 waves, one workgroup per CU (policy wgrad_ct: 0 auto, 128: A/B)
    // ... 64 or fewer (enc.conv2: 48 of a 128-channel tile's rows used): the 64-channel tile, 2 waves, three workgroups per CU
    const int ct = sc2_pol().wgrad_ct == 128 ? 128 : a.Cout > 128 ? 256 : a.Cout <= 64 ? 64 : 128;
    a.n_ctiles = (a.Cout + ct - 1) / ct;
    // enough pixel chunks to fill the chip several times over, at least 8 slabs each
    const long long tiles = (long long)a.n_ktiles * a.n_ctiles;
    // workgroups per launch = output tiles x pixel ranges.  Every workgroup ends with 128 x 128 f32 atomic adds, so the pixel split
    // is paid in atomics: 4 096 workgroups (16 per CU) -> 1 024 (two rounds of the 512 resident ones) took the small layers'
    // launches from 0.095 / 0.31 / 0.28 ms to 0.054 / 0.24 / 0.24 (the 512 x 512 gamma gradient stays at 0.89: it is bound by
    // the L2 -> LDS fill of its 128 x 128 tiles, 16 KB per 64 MFMAs), the training step + 2 %.  SC2_WGRAD_WGS overrides (A/B).
    // (round 5, ranges XCD-local -- tools/wgrad_times.py: four output tiles or fewer want 512 workgroups (igdn3's gamma 0.177 -> 0.152
    //  ms), a single 128-channel tile row with a long K -- enc.conv2, 19 k-tiles -- wants 4 096 (0.63 -> 0.53); the rest is flat
    //  between 1 024 and 2 048)
    // (256-channel tiles: one workgroup per CU; a single row of channel tiles -- dec.conv2 / dec.conv4 / igdn3's gamma -- wants one
    //  round of 256, the 512 x 512 gamma gradient 1 024: profiles/r05k_wgrad_times.txt)
    // (maps of a ResNet tail -- stage 2's trainable layer2 .. layer4, at most 2e5 pixels per batch of 256 -- want ONE round of
    //  workgroups: every workgroup pays its 128 x 128 / 256 x 128 atomic adds whatever its share of the pixels, and with 1 024 of them
    //  layer3's 256 -> 1024 gradient took 0.115 ms where 256 take 0.054: profiles/r05s_wgrad_head_times.txt)
    const bool small_m = M <= 262144;
    const int wg_auto = small_m ? (ct == 256 ? (tiles > 64 ? 512 : 256) : 512)
                        : ct == 256 ? ((tiles <= 2 || a.n_ctiles == 1) ? 256 : 1024) : ct == 64 ? 4096 : tiles <= 4 ? 512 : (a.n_ctiles == 1 ? 4096 : 1024);
    const int wg_target = sc2_pol().wgrad_wgs > 0 ? sc2_pol().wgrad_wgs : wg_auto;
    long long chunks = (wg_target + tiles - 1) / tiles;
    long long rows = (M + chunks - 1) / chunks;
    rows = (rows + WG_STAGES * WG_SLAB - 1) / (WG_STAGES * WG_SLAB) * (WG_STAGES * WG_SLAB);   // whole trips of the kernel's slab loop
    if (rows < 8 * WG_SLAB) rows = 8 * WG_SLAB;
    chunks = (M + rows - 1) / rows;
    chunks = (chunks + 7) / 8 * 8;      // the kernel deals the ranges to the eight XCDs
    a.rows_per_block = (int)rows;
    const long long grid = tiles * chunks;
    SC2_REQUIRE(grid > 0 && grid < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv2d_wgrad: grid out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)a.Cout * a.K * sizeof(float), s);
    SC2_REQUIRE(e == hipSuccess, SC2_ERR_LAUNCH, "conv2d_wgrad: memset failed: %s", hipGetErrorString(e));
    if (ct == 256) {
        constexpr size_t lds = (size_t)WG_STAGES * WgGeo<256>::STAGE;   // 96 KB
        static bool attr_set_dev[SC2_MAX_DEVICES] = {};
        bool &attr_set = attr_set_dev[sc2_device_slot()];
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_kernel<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_kernel<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (a.x_abs) hipLaunchKernelGGL((conv_wgrad_kernel<true, 256>), dim3((unsigned)grid), dim3(512), lds, s, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<false, 256>), dim3((unsigned)grid), dim3(512), lds, s, a);
    } else if (ct == 64) {
        constexpr size_t lds = (size_t)WG_STAGES * WgGeo<64>::STAGE;   // 48 KB
        if (a.x_abs) hipLaunchKernelGGL((conv_wgrad_kernel<true, 64>), dim3((unsigned)grid), dim3(128), lds, s, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<false, 64>), dim3((unsigned)grid), dim3(128), lds, s, a);
    } else {
        constexpr size_t lds = (size_t)WG_STAGES * WgGeo<128>::STAGE;   // 64 KB
        static bool attr_set_dev[SC2_MAX_DEVICES] = {};
        bool &attr_set = attr_set_dev[sc2_device_slot()];
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_kernel<true, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wgrad_kernel<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (a.x_abs) hipLaunchKernelGGL((conv_wgrad_kernel<true, 128>), dim3((unsigned)grid), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<false, 128>), dim3((unsigned)grid), dim3(256), lds, s, a);
    }
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
