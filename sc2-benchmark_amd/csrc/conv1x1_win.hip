// 1x1 convolution + bias (+ residual) (+ ReLU) for the ResNet tail's conv1 / conv3 / downsample layers (gfx950), in the
// structure of the window-plane 3x3 kernel (conv3x3_win.hip): torchvision Bottleneck.conv1 + bn1 + ReLU, conv3 + bn3 + identity
// + ReLU and downsample[0] + downsample[1] of layer2 / layer3 / layer4 in eval mode, the callers on the far side of the
// bottleneck path (sc2bench/models/backbone.py:235-254).
//     y[m, co] = act( sum_ci x[pix(m), ci] w[co, ci] + bias[co] (+ res[m, co]) ),   bf16 NHWC, m = flat output pixel.
//
// Why another 1x1 kernel.  The weights-in-registers kernels (conv1x1_stream.hip, conv1x1_kres.hip) win where a layer is
// HBM-bound with a short K.  The K >= 512 layers at 12 544 - 50 176 pixels (conv1 of layer3 / layer4, layer4's conv3) are not:
// 26 - 53 GFLOP per launch ran at 400 - 630 TFLOP/s with the matrix pipes 15 - 20 % busy (profiles/r02j_pmc_head_mfma_busy.txt),
// their time going into the resident-weight prologue (0.5 - 2 MB per workgroup), the K-half exchange and tile quantisation.
// Here, as in conv3x3_win.hip:
//   * a tile is 208 consecutive output pixels (13 MFMA row tiles, no image structure needed for a 1x1 layer) x 128 channels,
//     wave w = 32 channels x 13 row tiles (26 accumulator tiles), two workgroups per CU (Tile<13>; Tile<7> = 112 pixels,
//     three workgroups per CU, is an option that measured slower here);
//   * the pixel operand is staged per 64-channel slab as eight 16-byte-chunk planes [chunk][pixel][16 B] in a ring of two
//     (2 x 32 KB) or four slabs: fragment rows are one address register per row tile + immediates, no vector ALU in the K loop;
//   * weights go L2 -> registers, fragment-major, four k-steps ahead; ONE barrier per slab (52 MFMAs per wave);
//   * the weight rows are permuted at packing time so that a lane holds eight consecutive channels of a pixel: 13 sixteen-byte
//     stores (and residual loads) per lane, bias / residual / ReLU in registers.
// Stride 2 (the downsample layers) only changes which input pixel a window row is filled from.
#include <stdlib.h>

#include <type_traits>

#include "sc2_common.h"

#ifndef SC2_NT_WIN1
#define SC2_NT_WIN1 0   // non-temporal output stores: measured SLOWER here (the consumer launch finds part of this map in L2 / the memory-side cache: head + 2.5 %, dec.conv2 + 2 %); 1: A/B
#endif

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, lds_ptr_t dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// (masked lanes out of range instead of a branch around the store; literal soffset 0: see conv3x3_win.hip buf_store16_z)
typedef unsigned win_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store16_z(buf_rsrc_t r, uint32_t voff, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(win_u32x4_t{v.x, v.y, v.z, v.w}, r, (int)voff, 0, SC2_NT_WIN1 ? SC2_BUF_AUX_NT : 0);
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t, uint32_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void buf_store16_z(buf_rsrc_t, uint32_t, uint4) {}
#endif

// Weight fragments are loaded by INLINE ASM and waited for with hand-counted `s_waitcnt vmcnt(N)` (round 4; conv2x2_win.hip has
// the full note): with the compiler's own loads the conditional window fill in front of a slab made its scoreboard merge
// conservative -- the first k-step of every slab waited `vmcnt(9)` right behind eight freshly issued window pieces, i.e. for one of
// THEM to land (tools/audit_vmcnt.py): ~a memory round trip per 50 MFMAs.  The rules that keep this safe are in conv2x2_win.hip: a fragment
// register is read only by the MFMAs of its k-step, each of which consumes a pixel fragment that went through wait_lgkm behind
// wait_vm; the load of k-step k + PF goes into the registers of k-step k behind that step's last MFMA.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t rsrc_words(const void *base, uint32_t bytes) {   // raw buffer descriptor: base, no stride, size, 32-bit raw data format
    const uint64_t a = (uint64_t)(uintptr_t)base;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// ("s_nop 4": the hazard recognizer does not look into inline asm.  The scalar offset / descriptor may have been written by a
//  VALU instruction just before -- hipcc restores spilled SGPRs with v_readlane_b32 -- and a VMEM instruction needs 5 wait states
//  behind a VALU write of an SGPR it reads: without them the loads of the tail-mode kernel used a stale offset (wrong weights) and
//  conv1x1_win a stale descriptor (memory fault).)
__device__ __forceinline__ void wload16(u32x4_t &d, i32x4_t r, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen ; wfrag" : "=&v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {   // (does not name the fragment registers: conv2x2_win.hip, rule (ii))
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ u32x4_t lds_read16_imm(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4_t &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

struct P1Args {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, W, Cin]
    const uint16_t *__restrict__ w;      // bf16 [Cin/32][Cout/16][64][8]  (hip.pack_conv_win of the [Cout, Cin, 1, 1] weight)
    const float *__restrict__ bias;      // f32 [Cout]
    const uint16_t *__restrict__ res;    // bf16 [M, Cout] or null
    const uint16_t *__restrict__ mask;   // bf16 [M, Cout] or null: y = mask > 0 ? acc + bias [+ res] : 0 (gradient through a ReLU whose OUTPUT is `mask`)
    uint16_t *__restrict__ y;            // bf16 [M, Cout], M = N * OH * OW
    int N, H, W, OH, OW, stride, Cin, Cout, relu;
    int n_chunks;                        // Cout / 128
    long long M;
    unsigned x_bytes, w_bytes, y_bytes;
};

// MT_ MFMA row tiles per tile: 13 (208 pixels, two workgroups per CU or one with the 4-deep ring) or 7 (112 pixels, ~150 VGPRs,
// three workgroups per CU: twice as many, smaller workgroups, which lose less when slots are taken by kernels running beside them)
template <int MT_>
struct Tile {
    static constexpr int MT = MT_;
    static constexpr int PX = MT * 16;                    // output pixels per tile
    static constexpr int NRG = (PX + 63) / 64;            // 64-row direct-to-LDS pieces per plane
    static constexpr int PLANE = NRG * 1024;              // bytes per 16-byte-chunk plane
    static constexpr int WIN_BYTES = 8 * PLANE;           // one 64-channel slab = eight planes; NBUF of them form the ring
};
constexpr int PF = 4;                    // weight fragments are fetched this many k-steps ahead (4 k-steps per loop trip)

// one k-step: 32 channels = chunk planes 4 KS .. 4 KS + 3 of window PAR
template <class T, int I>
__device__ __forceinline__ void mma_chain(f32x4_t (&acc)[T::MT][2], u32x4_t (&av)[T::MT], const bf16x8_t &bf0, const bf16x8_t &bf1) {
    if constexpr (I < T::MT) {
        wait_lgkm<T::MT - 1 - I>(av[I]);
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[I]);
        acc[I][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, af, acc[I][0], 0, 0, 0);
        acc[I][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, af, acc[I][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_chain<T, I + 1>(acc, av, bf0, bf1);
    }
}

// NVM: vmcnt budget of the wait for b0 / b1 (asm loads)
template <class T, int PAR, int KS, int NVM>
__device__ __forceinline__ void k_step(f32x4_t (&acc)[T::MT][2], const uint32_t (&a_base)[T::MT], u32x4_t &b0, u32x4_t &b1) {
    constexpr int MT = T::MT;
    constexpr int OFF = (PAR & 1) * T::WIN_BYTES + KS * 4 * T::PLANE;   // (a_base points at window PAR & ~1: 16-bit immediates)
    u32x4_t av[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) av[i] = lds_read16_imm<OFF>(a_base[i]);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<NVM>();
    const bf16x8_t bf0 = __builtin_bit_cast(bf16x8_t, b0), bf1 = __builtin_bit_cast(bf16x8_t, b1);
    mma_chain<T, 0>(acc, av, bf0, bf1);
}

// NBUF = 2: 64 KB of LDS, two workgroups per CU (the partner covers the slab that is not yet there): layers with many tiles.
// NBUF = 4: 128 KB, one workgroup per CU, three slabs (1.5 us of MFMA work) in flight: the 12 544-pixel layers of layer4, whose
// 244 - 976 workgroups would otherwise wait for every slab (one slab ahead = 0.5 us against ~2 us of loaded HBM latency).
template <int NBUF, class T>
__global__ __launch_bounds__(256, NBUF == 2 ? (T::MT == 7 ? 3 : 2) : 1) void conv1x1_win_kernel(const P1Args p) {
    constexpr int MT = T::MT, PX = T::PX, NRG = T::NRG, PLANE = T::PLANE, WIN_BYTES = T::WIN_BYTES;
    constexpr uint32_t OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cin = p.Cin, Cout = p.Cout;
    const int NS = Cin >> 6;   // 64-channel slabs

    // XCD x gets a contiguous range of (pixel tile, channel chunk) pairs, chunk fastest: the chunks of one pixel tile read
    // their window through the same L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int chunk = bid % p.n_chunks, mtile = bid / p.n_chunks;
    const int n0 = chunk * 128 + wave * 32;
    const long long m0 = (long long)mtile * PX;

    const i32x4_t rs_w = rsrc_words(p.w, p.w_bytes);

    // window fill: wave w fills chunk planes 2 w and 2 w + 1 (chunk c = channels [8 c, 8 c + 8) of the slab); piece j = rows
    // [64 j, 64 j + 64) = output pixels m0 + 64 j + lane.  The per-lane source offsets stay in registers.
    uint32_t pw_vo[NRG];
    // (stride tested once, lane masks as selects: branch-free)
    if (p.stride != 1) {
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < NRG; ++j) {
            const long long m = m0 + j * 64 + lane;
            const bool ok = (j * 64 + lane < PX) & (m < p.M);
            const long long mc = ok ? m : 0;
            const int n = (int)(mc / ohw), rem = (int)(mc - (long long)n * ohw);
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const long long pix = ((long long)n * p.H + oh * p.stride) * p.W + ow * p.stride;
            pw_vo[j] = ok ? (uint32_t)(pix * Cin * 2) : OOB;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NRG; ++j) {
            const long long m = m0 + j * 64 + lane;
            const bool ok = (j * 64 + lane < PX) & (m < p.M);
            pw_vo[j] = ok ? (uint32_t)(m * Cin * 2) : OOB;
        }
    }
    // (EVERY slab start issues its 2 NRG pieces -- the k-steps' vmcnt budgets count them: past the last slab they come from a
    //  zero-sized descriptor, i.e. zeros into a dead buffer)
    auto issue_window = [&](int cb, int par) {
        const buf_rsrc_t rs_x = make_rsrc(p.x, cb < NS ? p.x_bytes : 0u);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < NRG; ++j)
                buf_load_lds16(rs_x, (lds_ptr_t)(smem + par * WIN_BYTES + (wave * 2 + c) * PLANE + j * 1024), pw_vo[j],
                               (uint32_t)cb * 128u + (uint32_t)(wave * 2 + c) * 16u);
        }
    };

    // fragment rows of this lane: k-lanes fq = chunk fq of the k-step's four planes
    uint32_t a_base[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_base[i] = lds_base + (uint32_t)(fq * PLANE + (i * 16 + frow) * 16);
    uint32_t a_base2[MT];   // windows 2, 3 of the four-deep ring
#pragma unroll
    for (int i = 0; i < MT; ++i) a_base2[i] = a_base[i] + 2u * WIN_BYTES;

    // weights: k-step kt, 16-channel tile t -> 1 KB at ((kt * Cout/16) + t) * 1024; this wave's tiles are n0/16, n0/16 + 1
    const uint32_t b_vo = (uint32_t)(lane * 16);
    const uint32_t b_step = (uint32_t)(Cout >> 4) * 1024u;
    const uint32_t KT = (uint32_t)NS * 2u;
    const uint32_t b_so0 = (uint32_t)(n0 >> 4) * 1024u;
    auto fetch_b = [&](uint32_t kt, u32x4_t &b0, u32x4_t &b1) {   // (past the end: the last k-step again, never used)
        const uint32_t so = b_so0 + (kt < KT - 1u ? kt : KT - 1u) * b_step;
        wload16(b0, rs_w, b_vo, so);
        wload16(b1, rs_w, b_vo, so + 1024u);
    };

    f32x4_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) issue_window(d, d);
    u32x4_t bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch_b((uint32_t)s, bq[s][0], bq[s][1]);

    // a k-step's fragments were fetched PF k-steps = two slabs ago: behind them the 2 (PF - 1) loads of the three steps in
    // between and the window pieces of two slab starts; the fetch of k-step + PF goes into the same registers behind the step's
    // last MFMA
    constexpr int VM_STEP = 2 * (PF - 1) + 2 * (2 * NRG);
#define SC2_P1_STEP(PAR, cb, KS, SLOT)                                                 \
    {                                                                                 \
        if constexpr ((PAR) < 2) k_step<T, PAR, KS, VM_STEP>(acc, a_base, bq[SLOT][0], bq[SLOT][1]); \
        else k_step<T, PAR, KS, VM_STEP>(acc, a_base2, bq[SLOT][0], bq[SLOT][1]);     \
        fetch_b((uint32_t)(cb) * 2u + (KS + PF), bq[SLOT][0], bq[SLOT][1]);           \
    }
    // this wave's share of window cb has landed when at most the loads issued after it are outstanding: the NBUF - 2 younger
    // windows (8 pieces each) and the 2 x 2 weight fetches of each of the NBUF - 1 slabs since (fewer near the ends: a
    // conservative wait)
    constexpr int YOUNGER = (NBUF - 2) * (2 * NRG) + (NBUF - 1) * 4;
#define SC2_P1_SLAB(PAR, cb, SLOT0)                                                                     \
    {                                                                                                   \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");                                   \
        __builtin_amdgcn_s_barrier();   /* window cb complete; everybody is done with window cb - 1 */   \
        issue_window((cb) + NBUF - 1, (PAR + NBUF - 1) % NBUF);                                          \
        SC2_P1_STEP(PAR, cb, 0, SLOT0) SC2_P1_STEP(PAR, cb, 1, SLOT0 + 1)                                \
    }
    // (the windows of the prologue have fewer weight fetches behind them than YOUNGER assumes: drain everything once)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NBUF == 2) {
        for (int cb = 0; cb < NS; cb += 2) {
            SC2_P1_SLAB(0, cb, 0)
            SC2_P1_SLAB(1, cb + 1, 2)
        }
    } else {
        for (int cb = 0; cb < NS; cb += 4) {
            SC2_P1_SLAB(0, cb, 0)
            SC2_P1_SLAB(1, cb + 1, 2)
            SC2_P1_SLAB(2, cb + 2, 0)
            SC2_P1_SLAB(3, cb + 3, 2)
        }
    }
#undef SC2_P1_SLAB
#undef SC2_P1_STEP
    // The last PF k-steps fetched "the last k-step again" (never used): those loads are still in flight here, and the compiler
    // regards their registers as dead -- epilogue values computed into them were overwritten when the loads landed (memory
    // access faults at bs 256, where they land late).  The wait names the registers, which keeps them allocated up to it.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]), "+v"(bq[3][0]),
                   "+v"(bq[3][1])::"memory");

    // epilogue: lane (frow, fq) holds, for row tile i, output channels n0 + 8 fq + [0, 4) in acc[i][0] and + [4, 8) in acc[i][1]
    // of pixel m0 + i * 16 + frow
    const float4 bias_lo = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq);
    const float4 bias_hi = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq + 4);
    // Branch-free (round 4): the flags are tested once, masked lanes load / store out of range through descriptors (a load out
    // of range returns zeros) -- as per-row-tile `if`s the epilogue of a 3 - 6 us workgroup held ~75 scalar branches.
    const buf_rsrc_t rs_y = make_rsrc(p.y, p.y_bytes);
    const buf_rsrc_t rs_r = make_rsrc(p.res ? p.res : p.y, p.res ? p.y_bytes : 0u);
    const buf_rsrc_t rs_m = make_rsrc(p.mask ? p.mask : p.y, p.mask ? p.y_bytes : 0u);
    auto finish = [&](auto relu_c, auto res_c, auto mask_c) {
        constexpr bool RELU = decltype(relu_c)::value, HAS_RES = decltype(res_c)::value, MASK = decltype(mask_c)::value;
        uint4 rv[MT];
        if (HAS_RES) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const long long m = m0 + i * 16 + frow;
                rv[i] = buf_load16(rs_r, m < p.M ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const long long m = m0 + i * 16 + frow;
            float v[8] = {acc[i][0][0] + bias_lo.x, acc[i][0][1] + bias_lo.y, acc[i][0][2] + bias_lo.z, acc[i][0][3] + bias_lo.w,
                          acc[i][1][0] + bias_hi.x, acc[i][1][1] + bias_hi.y, acc[i][1][2] + bias_hi.z, acc[i][1][3] + bias_hi.w};
            if (HAS_RES) {
                v[0] += bf_lo(rv[i].x); v[1] += bf_hi(rv[i].x); v[2] += bf_lo(rv[i].y); v[3] += bf_hi(rv[i].y);
                v[4] += bf_lo(rv[i].z); v[5] += bf_hi(rv[i].z); v[6] += bf_lo(rv[i].w); v[7] += bf_hi(rv[i].w);
            }
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (MASK) {   // (round 5: the ReLU-gradient pass over this tensor folded in; loaded per row tile behind the arithmetic)
                const uint4 mq = buf_load16(rs_m, m < p.M ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, 0u);
                const uint32_t mw[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = bf_lo(mw[e]) > 0.f ? v[2 * e] : 0.f;
                    v[2 * e + 1] = bf_hi(mw[e]) > 0.f ? v[2 * e + 1] : 0.f;
                }
            }
            const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            buf_store16_z(rs_y, m < p.M ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, o);
        }
    };
    if (p.mask != nullptr) {
        if (p.res != nullptr) finish(std::false_type{}, std::true_type{}, std::true_type{});
        else finish(std::false_type{}, std::false_type{}, std::true_type{});
    } else if (p.res != nullptr) {
        if (p.relu != 0) finish(std::true_type{}, std::true_type{}, std::false_type{});
        else finish(std::false_type{}, std::true_type{}, std::false_type{});
    } else {
        if (p.relu != 0) finish(std::true_type{}, std::false_type{}, std::false_type{});
        else finish(std::false_type{}, std::false_type{}, std::false_type{});
    }
}

template <int NBUF, class T>
int launch_p1(const P1Args &a, hipStream_t s) {
    constexpr int LDS_BYTES = NBUF * T::WIN_BYTES;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1x1_win_kernel<NBUF, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_set = true;
    }
    const long long n_wg = (a.M + T::PX - 1) / T::PX * a.n_chunks;
    SC2_REQUIRE(n_wg < (1ll << 31), SC2_ERR_UNSUPPORTED, "conv1x1_win: grid too large");
    hipLaunchKernelGGL((conv1x1_win_kernel<NBUF, T>), dim3((unsigned)n_wg), dim3(256), LDS_BYTES, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv1x1_win_supported(int Cin, int Cout, int stride) {
    return (stride == 1 || stride == 2) && Cin >= 128 && Cin % 128 == 0 && Cout >= 128 && Cout % 128 == 0 ? 1 : 0;
}

extern "C" int sc2_conv1x1_win_fwd(const void *x, const void *w_frag, const float *bias, const void *residual, const void *mask, void *y,
                                   int N, int H, int W, int Cin, int Cout, int stride, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv1x1_win: null argument");
    SC2_REQUIRE(!(mask && relu), SC2_ERR_INVALID_ARG, "conv1x1_win: mask and relu are exclusive");
    SC2_REQUIRE(N > 0 && H > 0 && W > 0, SC2_ERR_INVALID_ARG, "conv1x1_win: non-positive shape");
    SC2_REQUIRE(sc2_conv1x1_win_supported(Cin, Cout, stride), SC2_ERR_UNSUPPORTED,
                "conv1x1_win: needs Cin %% 128 == 0, Cout %% 128 == 0, stride 1 or 2 (got %d -> %d, stride %d)", Cin, Cout, stride);
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const long long M = (long long)N * OH * OW;
    const long long x_bytes = (long long)N * H * W * Cin * 2, w_bytes = (long long)Cin * Cout * 2, y_bytes = M * Cout * 2;
    SC2_REQUIRE(x_bytes < 0x7FF00000LL && w_bytes < 0x7FF00000LL && y_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED,
                "conv1x1_win: operand of %lld bytes exceeds 2 GB", x_bytes > y_bytes ? x_bytes : y_bytes);
    P1Args a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.res = static_cast<const uint16_t *>(residual);
    a.mask = static_cast<const uint16_t *>(mask);
    a.y = static_cast<uint16_t *>(y);
    a.N = N; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.stride = stride; a.Cin = Cin; a.Cout = Cout; a.relu = relu ? 1 : 0;
    a.n_chunks = Cout / 128;
    a.M = M;
    a.x_bytes = (unsigned)x_bytes; a.w_bytes = (unsigned)w_bytes; a.y_bytes = (unsigned)y_bytes;
    const long long n_wg = (M + 207) / 208 * a.n_chunks;   // (workgroups of the 208-pixel tiling)
    // ring depth: four buffers (one workgroup per CU) when the launch has about one workgroup per CU anyway (layer4's conv1 at
    // bs 256: 244 workgroups, 0.050 -> 0.045 ms; every launch with more workgroups measured slower that way) and K is a
    // multiple of 256; SC2_P1_NBUF = 2 | 4 overrides (A/B)
    int nbuf = (Cin % 256 == 0 && n_wg <= 320) ? 4 : 2;
    if (const int v = sc2_pol().p1_nbuf) {
        if (v == 2 || (v == 4 && Cin % 256 == 0)) nbuf = v;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nbuf == 4) return launch_p1<4, Tile<13>>(a, s);
    // half tiles (SC2_P1_HALF=1; off by default): 112-pixel tiles, three workgroups per CU.  Unlike the 3x3 kernels (conv3x3_win.hip)
    // this one loses with them -- conv1 of layer2 0.068 -> 0.076 ms, bench - 0.3 %: per pixel a 1x1 layer streams twice the weights
    const int half = sc2_pol().p1_half;
    return half ? launch_p1<2, Tile<7>>(a, s) : launch_p1<2, Tile<13>>(a, s);
}
