// 3x3 stride-1 pad-1 convolution + bias (+ ReLU) for the ResNet tail's conv2 layers (gfx950): torchvision
// Bottleneck.conv2 + bn2 + ReLU of layer2 / layer3 / layer4 in eval mode, the callers on the far side of the bottleneck
// path (sc2bench/models/backbone.py:235-254), at the 28 / 14 / 7 pixel maps a 224 x 224 input reaches them with.
//     y[n, oh, ow, co] = act( sum_{kh, kw, ci} x[n, oh + kh - 1, ow + kw - 1, ci] w[co, ci, kh, kw] + bias[co] ),  bf16 NHWC.
//
// Why a dedicated kernel.  On the window-staged tile kernel (conv_igemm_impl.h, Cfg8::PATCH3) these ten launches ran at
// 480 - 620 TFLOP/s: 256-pixel tiles quantise 196-pixel images badly (392 tiles on 512 slots), every fragment read pays
// ~8 vector instructions of swizzled, image-clipped address arithmetic, and every 16 MFMAs pay a barrier pair.  Here
//   * a tile is 196 output pixels (one 14 x 14 image, seven rows of a 28 x 28 image, four 7 x 7 images = 13 MFMA row tiles)
//     x 128 output channels: 512 or 1024 equal workgroups for bs 256, two per CU;
//   * per 32-channel slab the tile's input window is staged ONCE, ZERO-PADDED (the halo is physically in LDS: no clipping
//     in the loop), as four 16-byte-chunk PLANES [chunk][window row][16 B].  In that layout the fragment rows of a lane
//     for tap (kh, kw) sit at a constant byte distance (kh (W + 2) + kw) * 16 from those of tap (0, 0): the nine taps of a
//     slab read through ONE address register per row tile and nine immediate offsets - no vector ALU in the K loop;
//   * wave w owns output channels [32 w, 32 w + 32) of the tile for all 13 row tiles (26 accumulator tiles).  Its weight
//     fragments are nobody else's, so they never touch LDS: two buffer_load_dwordx4 per 26 MFMAs, straight from L2 into
//     registers, three k-steps ahead (fragment-major packing: one operand = 1 KB contiguous);
//   * the only shared operand is the window, so the waves meet at ONE barrier per slab = per 234 MFMAs per wave;
//   * the weight rows are permuted at packing time so that a lane ends up with EIGHT consecutive output channels of a
//     pixel: 13 sixteen-byte stores per lane, bias + ReLU in registers, no LDS staging.
// The two workgroups of a CU are independent: while one wave of a SIMD reads its 13 pixel fragments the other issues MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "sc2_common.h"

#ifndef SC2_NT_WIN3
#define SC2_NT_WIN3 0   // non-temporal output stores: measured SLOWER here (the consumer launch finds part of this map in L2 / the memory-side cache: head + 2.5 %, dec.conv2 + 2 %); 1: A/B
#endif

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}

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
// 16-byte store through a descriptor, masked lanes sent out of range instead of branching around the store.  soffset is the
// LITERAL 0: that is the form for which hipcc inserts the wait states of the ">64-bit store data" hazard itself (with an SGPR
// soffset it does not: conv2x2_win.hip buf_store16, tools/micro/store_hazard.hip)
typedef unsigned win_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store16_z(buf_rsrc_t r, uint32_t voff, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(win_u32x4_t{v.x, v.y, v.z, v.w}, r, (int)voff, 0, SC2_NT_WIN3 ? SC2_BUF_AUX_NT : 0);
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t, uint32_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void buf_store16_z(buf_rsrc_t, uint32_t, uint4) {}
#endif

// Weight fragments of the stride-1 kernel are loaded by INLINE ASM and waited for with hand-counted `s_waitcnt vmcnt(N)` (round 4;
// conv2x2_win.hip has the full note): with the compiler's own loads the conditional window fill in front of a slab made its
// scoreboard merge conservative, and the first k-step of every other slab waited `vmcnt(2)` = for the window pieces of the NEXT
// slab issued a few instructions earlier (tools/audit_vmcnt.py).  The rules that keep this safe are in conv2x2_win.hip: a fragment
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
// wait until at most N of this wave's LDS reads are outstanding; `v` (the destination of the read being waited for) is
// threaded through so that its consumers cannot be scheduled above the wait
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4_t &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

struct WinArgs {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, W, Cin]
    const uint16_t *__restrict__ w;      // bf16 [Cin/32 * 9][Cout/16][64][8]  (sc2_conv3x3_win_fwd's packing)
    const float *__restrict__ bias;      // f32 [Cout]
    uint16_t *__restrict__ y;            // bf16 NHWC [N, H, W, Cout]
    const uint16_t *__restrict__ mask;   // bf16 like y or null: y = mask > 0 ? acc + bias : 0 (the gradient through a ReLU whose OUTPUT is `mask`)
    int N, Cin, Cout, relu;
    int n_chunks;                        // Cout / 128
    int n_mtiles;
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned long long *stamps;          // DEBUG (SC2_WIN_STAMPS=1): [workgroup][start, end] of the 100 MHz wall clock, or null
};

// W: map width = height; ROWS: output rows of one image per tile; IMGS: images per tile (ROWS == W when > 1)
template <int W_, int ROWS_, int IMGS_, int PWD_ = W_ + 2>
struct Geo {
    static constexpr int W = W_, H = W_, ROWS = ROWS_, IMGS = IMGS_;
    static constexpr int TILES_PER_IMG = H / ROWS;               // (IMGS == 1)
    static constexpr int PWD = PWD_;                             // window row pitch (>= W + 2)
    static constexpr int IMGP = (ROWS + 2) * PWD;                // window rows per image
    static constexpr int WROWS = IMGS * IMGP;
    static constexpr int NRG = (WROWS + 63) / 64;                // 64-row direct-to-LDS pieces per plane
    static constexpr int PLANE = NRG * 64 * 16;                  // bytes per chunk plane (a multiple of 256: bank-aligned planes)
    static constexpr int WIN_BYTES = 4 * PLANE;
    static constexpr int PX = IMGS * ROWS * W;                   // output pixels per tile
    static constexpr int MT = (PX + 15) / 16;
    static constexpr int LDS_BYTES = 2 * WIN_BYTES;
    static_assert(H % ROWS == 0 && (IMGS == 1 || ROWS == H), "whole rows of one image, or whole images");
    static_assert(PLANE % 256 == 0 && (MT == 13 || MT == 7), "13 (or, half tiles, 7) row tiles, bank-aligned planes");
    static_assert(WIN_BYTES + (2 * PWD + 2) * 16 < 65536, "tap offsets are 16-bit immediates");
    static constexpr int tap_off(int tap) { return ((tap / 3) * PWD + tap % 3) * 16; }   // bytes from tap (0, 0)'s fragment row
};

// Stride 2 (pad 1): the input pixel of output (oh, ow) at tap (kh, kw) is (2 oh + kh - 1, 2 ow + kw - 1).  In padded window
// coordinates (ihp, iwp) = (2 ohl + kh, 2 ow + kw) the PARITY of a position is a property of the tap alone, so the window is
// stored as four parity classes [(ihp & 1, iwp & 1)][ihp >> 1][iwp >> 1]: inside a class, neighbouring output pixels are
// neighbouring window rows again and tap (kh, kw) is the constant distance
//     ((kh & 1) * 2 + (kw & 1)) * CLS + (kh >> 1) * PWD + (kw >> 1)      window rows
// from tap (0, 0) -- immediates, as for stride 1.  OW: OUTPUT map width = height (the input map is 2 OW wide).
template <int OW_, int ROWS_, int IMGS_>
struct GeoS2 {
    static constexpr int OW = OW_, OH = OW_, W = 2 * OW_, H = 2 * OW_, ROWS = ROWS_, IMGS = IMGS_;
    static constexpr int TILES_PER_IMG = OH / ROWS;
    static constexpr int PWD = OW + 1;                           // class row pitch (iwp >> 1 in [0, OW])
    static constexpr int CLS = (ROWS + 1) * PWD;                 // window rows per parity class (ihp >> 1 in [0, ROWS])
    static constexpr int IMGP = 4 * CLS;
    static constexpr int WROWS = IMGS * IMGP;
    static constexpr int NRG = (WROWS + 63) / 64;
    static constexpr int PLANE = NRG * 64 * 16;
    static constexpr int WIN_BYTES = 4 * PLANE;
    static constexpr int PX = IMGS * ROWS * OW;
    static constexpr int MT = (PX + 15) / 16;
    static constexpr int LDS_BYTES = WIN_BYTES;                  // ONE window (60 - 64 KB): two workgroups per CU
    static_assert(OH % ROWS == 0 && (IMGS == 1 || ROWS == OH), "whole rows of one image, or whole images");
    static_assert(PLANE % 256 == 0 && (MT == 13 || MT == 7), "13 (or, half tiles, 7) row tiles, bank-aligned planes");
    static_assert(4 * CLS * 16 < 65536, "tap offsets are 16-bit immediates");
    static constexpr int tap_off(int tap) {
        const int kh = tap / 3, kw = tap % 3;
        return (((kh & 1) * 2 + (kw & 1)) * CLS + (kh >> 1) * PWD + (kw >> 1)) * 16;
    }
};

constexpr int PF = 3;   // weight fragments are fetched this many k-steps ahead (18 k-steps per loop trip: 18 % PF == 0)

#ifndef SC2_W3_ORDER
#define SC2_W3_ORDER 0   // 1: operand-stationary MFMA order inside a k-step (every row tile with b0, then every row tile with b1): measured 1.3 % SLOWER here (the five 3x3 shapes 0.3255 - 0.3303 -> 0.3298 - 0.3387 ms, profiles/r06y_mfma_order_ab.txt) although it gains 1 % in conv2x2_win; 0 (default): b0 / b1 alternate
#endif
// MFMAs of one k-step, row tile I onwards: each waits for its own fragment read only (MT - 1 - I younger reads may be in flight).
// Round 6 A/B (SC2_W3_ORDER = 1, off): one weight fragment for MT MFMAs in a row, then the other -- the same products into the same accumulators in
// the same k order (bit-identical); an operand that does not change between MFMAs toggles less, and these launches are power-limited
// (conv2x2_win.hip, mma_step; tools/clock_probe.py --zero-data).
template <class G, int DBG, int I, int PASS = 0>
__device__ __forceinline__ void mma_chain(f32x4_t (&acc)[G::MT][2], u32x4_t (&av)[G::MT], const bf16x8_t &bf0, const bf16x8_t &bf1) {
#if SC2_W3_ORDER
    if constexpr (I < G::MT) {
        if constexpr (PASS == 0 && !(DBG & 4)) wait_lgkm<G::MT - 1 - I>(av[I]);
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[I]);
        if constexpr (PASS == 0) acc[I][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, af, acc[I][0], 0, 0, 0);
        else acc[I][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, af, acc[I][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_chain<G, DBG, I + 1, PASS>(acc, av, bf0, bf1);
    } else if constexpr (PASS == 0) {
        mma_chain<G, DBG, 0, 1>(acc, av, bf0, bf1);
    }
#else
    if constexpr (I < G::MT) {
        if constexpr (!(DBG & 4)) wait_lgkm<G::MT - 1 - I>(av[I]);
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[I]);
        acc[I][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, af, acc[I][0], 0, 0, 0);
        acc[I][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, af, acc[I][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_chain<G, DBG, I + 1>(acc, av, bf0, bf1);
    }
#endif
}

// NVM >= 0: b0 / b1 come from asm loads (wload16): wait until at most NVM younger vector-memory operations are outstanding
template <class G, int PAR, int TAP, int DBG, int NVM = -1, class B>
__device__ __forceinline__ void k_step(f32x4_t (&acc)[G::MT][2], const uint32_t (&a_base)[G::MT], B &b0, B &b1) {
    constexpr int MT = G::MT;
    constexpr int OFF = PAR * G::WIN_BYTES + G::tap_off(TAP);
    u32x4_t av[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if constexpr (DBG & 4) av[i] = u32x4_t{a_base[i], a_base[i], a_base[i], a_base[i]};   // timing experiment: no fragment reads
        else av[i] = lds_read16_imm<OFF>(a_base[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NVM >= 0) wait_vm<NVM>();
    const bf16x8_t bf0 = __builtin_bit_cast(bf16x8_t, b0), bf1 = __builtin_bit_cast(bf16x8_t, b1);
    mma_chain<G, DBG, 0>(acc, av, bf0, bf1);
}

// DBG (timing experiments, results garbage): 1 no window refills, 2 no weight fetches in the loop, 4 no fragment reads, 8 no barriers
// MASK_: the epilogue's ReLU-gradient form (y = mask > 0 ? value : 0) is an instantiation of its own -- as a runtime case it cost the
// half-tile kernels of the inference head ten registers (128 -> 138: one workgroup per CU less)
template <class G, int DBG = 0, bool MASK_ = false>
__global__ __launch_bounds__(256, G::MT == 7 ? 3 : 2) void conv3x3_win_kernel(const WinArgs p) {
    constexpr int MT = G::MT, W = G::W, H = G::H;
    constexpr uint32_t OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    if (p.stamps && tid == 0) p.stamps[2 * blockIdx.x] = wall_clock64();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cin = p.Cin, Cout = p.Cout;
    const int NS = Cin >> 5;   // 32-channel slabs

    // XCD x gets a contiguous range of (row tile, channel chunk) pairs, chunk fastest: the chunks of one row tile read
    // their window through the same L2.  (One chunk per XCD, which keeps layer4's 4.7 MB of weights within every 4 MB L2,
    // measured the same.)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int chunk = bid % p.n_chunks, mtile = bid / p.n_chunks;
    const int n0 = chunk * 128 + wave * 32;
    // first image / first output row of the tile
    const int img0 = G::IMGS > 1 ? mtile * G::IMGS : mtile / G::TILES_PER_IMG;
    const int row0 = G::IMGS > 1 ? 0 : (mtile % G::TILES_PER_IMG) * G::ROWS;

    const i32x4_t rs_w = rsrc_words(p.w, p.w_bytes);

    // window fill: wave w fills plane w (chunk w of every row); piece j = window rows [64 j, 64 j + 64).  The per-lane
    // source offsets stay in registers: recomputing them per slab (~25 vector instructions per piece) cost 10 - 20 % of
    // the kernel -- every vector instruction competes with the MFMAs for the SIMD's issue port
    uint32_t pw_vo[G::NRG];
#pragma unroll
    for (int j = 0; j < G::NRG; ++j) {
        const int wr = j * 64 + lane;
        const int il = wr / G::IMGP, rem = wr - il * G::IMGP;
        const int ihp = rem / G::PWD, iwp = rem - ihp * G::PWD;
        const int img = img0 + il, ih = row0 + ihp - 1, iw = iwp - 1;
        const bool ok = (wr < G::WROWS) & (img < p.N) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
        pw_vo[j] = ok ? (uint32_t)((((img * H + ih) * W + iw) * Cin) * 2 + wave * 16) : OOB;
    }
    // (EVERY slab start issues its NRG pieces -- the k-steps' vmcnt budgets count them: behind the last slab they come from a
    //  zero-sized descriptor, i.e. zeros into the dead buffer)
    auto issue_window = [&](int cb, int par) {
        const buf_rsrc_t rs_x = make_rsrc(p.x, cb < NS ? p.x_bytes : 0u);
#pragma unroll
        for (int j = 0; j < G::NRG; ++j)
            buf_load_lds16(rs_x, (lds_ptr_t)(smem + par * G::WIN_BYTES + wave * G::PLANE + j * 1024), pw_vo[j], (uint32_t)cb * 64u);
    };

    // fragment rows of this lane at tap (0, 0)
    uint32_t a_base[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = i * 16 + frow;
        m = m < G::PX ? m : G::PX - 1;   // (rows past the tile: any valid address, results discarded)
        const int il = m / (G::ROWS * W), rem = m - il * (G::ROWS * W);
        const int ohl = rem / W, ow = rem - ohl * W;
        a_base[i] = lds_base + (uint32_t)(fq * G::PLANE + (il * G::IMGP + ohl * G::PWD + ow) * 16);
    }

    // weights: k-step kt, 16-channel tile t -> 1 KB at ((kt * Cout/16) + t) * 1024; this wave's tiles are n0/16, n0/16 + 1
    const uint32_t b_vo = (uint32_t)(lane * 16);
    const uint32_t b_step = (uint32_t)(Cout >> 4) * 1024u;
    const uint32_t KT = (uint32_t)NS * 9u;
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

    issue_window(0, 0);
    u32x4_t bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch_b((uint32_t)s, bq[s][0], bq[s][1]);

    // k-step TAP: its fragments were fetched PF k-steps ago, behind them the 2 (PF - 1) loads of the steps in between and -- for
    // the first PF taps of a slab -- the NRG window pieces of this slab's start; the fetch of k-step + PF goes into the same
    // registers behind the step's last MFMA
#define SC2_WIN_STEP(PAR, cb, TAP, SLOT)                                              \
    {                                                                                 \
        k_step<G, PAR, TAP, DBG, (DBG & 3) ? 0 : 2 * (PF - 1) + (TAP < PF ? G::NRG : 0)>(acc, a_base, bq[SLOT][0], bq[SLOT][1]); \
        if constexpr (!(DBG & 2)) fetch_b((uint32_t)(cb) * 9u + (TAP + PF), bq[SLOT][0], bq[SLOT][1]); \
    }
#define SC2_WIN_SLAB(PAR, cb)                                                                           \
    {                                                                                                   \
        /* this wave's share of window cb has landed: it is older than the 2 PF weight loads in flight */ \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PF) : "memory");                                    \
        if constexpr (!(DBG & 8)) __builtin_amdgcn_s_barrier();   /* window cb complete; everybody is done with window cb - 1 */ \
        if constexpr (!(DBG & 1)) issue_window((cb) + 1, 1 - PAR);                                       \
        SC2_WIN_STEP(PAR, cb, 0, 0) SC2_WIN_STEP(PAR, cb, 1, 1) SC2_WIN_STEP(PAR, cb, 2, 2)              \
        SC2_WIN_STEP(PAR, cb, 3, 0) SC2_WIN_STEP(PAR, cb, 4, 1) SC2_WIN_STEP(PAR, cb, 5, 2)              \
        SC2_WIN_STEP(PAR, cb, 6, 0) SC2_WIN_STEP(PAR, cb, 7, 1) SC2_WIN_STEP(PAR, cb, 8, 2)              \
    }
    for (int cb = 0; cb < NS; cb += 2) {
        SC2_WIN_SLAB(0, cb)
        SC2_WIN_SLAB(1, cb + 1)
    }
#undef SC2_WIN_SLAB
#undef SC2_WIN_STEP
    // The last PF k-steps fetched "the last k-step again" (never used): those loads are still in flight here, and the compiler
    // regards their registers as dead -- the epilogue's store addresses computed into them were overwritten when the loads
    // landed (memory access faults at bs 256, where they land late).  The wait names the registers, which keeps them allocated
    // up to it.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1])::"memory");

    // epilogue: lane (frow, fq) holds, for row tile i, output channels n0 + 8 fq + [0, 4) in acc[i][0] and + [4, 8) in acc[i][1]
    // (the packing permutes the weight rows that way) of pixel i * 16 + frow
    const float4 bias_lo = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq);
    const float4 bias_hi = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq + 4);
    const long long m_base = G::IMGS > 1 ? (long long)img0 * (H * W) : ((long long)img0 * H + row0) * W;
    const long long M = (long long)p.N * H * W;
    // Branch-free (round 4): the ReLU flag is tested once (as `if (relu)` inside the unrolled loop it was two scalar branches per
    // row tile) and masked lanes store out of range through a descriptor instead of jumping around the store -- a workgroup
    // lives for 3 - 6 us, and its ~60 branches were a measurable part of that.
    const buf_rsrc_t rs_y = make_rsrc(p.y, p.y_bytes);
    [[maybe_unused]] const buf_rsrc_t rs_m = make_rsrc(MASK_ ? p.mask : p.y, MASK_ ? p.y_bytes : 0u);
    // MASK (round 5): the layer is the data gradient of a conv whose INPUT was a fused ReLU's output: y = mask > 0 ? value : 0 -- the
    // separate ReLU-gradient pass over this tensor (read, read, write) becomes one extra 16-byte load per row tile here
    auto finish = [&](auto relu_c, auto mask_c) {
        constexpr bool RELU = decltype(relu_c)::value, MASK = decltype(mask_c)::value;
        uint4 mv[MT];
        if (MASK) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ml = i * 16 + frow;
                const long long m = m_base + ml;
                mv[i] = buf_load16(rs_m, ((ml < G::PX) & (m < M)) ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ml = i * 16 + frow;
            const long long m = m_base + ml;
            float v[8] = {acc[i][0][0] + bias_lo.x, acc[i][0][1] + bias_lo.y, acc[i][0][2] + bias_lo.z, acc[i][0][3] + bias_lo.w,
                          acc[i][1][0] + bias_hi.x, acc[i][1][1] + bias_hi.y, acc[i][1][2] + bias_hi.z, acc[i][1][3] + bias_hi.w};
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (MASK) {
                const uint32_t mw[4] = {mv[i].x, mv[i].y, mv[i].z, mv[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __builtin_bit_cast(float, mw[e] << 16) > 0.f ? v[2 * e] : 0.f;
                    v[2 * e + 1] = __builtin_bit_cast(float, mw[e] & 0xFFFF0000u) > 0.f ? v[2 * e + 1] : 0.f;
                }
            }
            const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            const bool ok = (ml < G::PX) & (m < M);
            buf_store16_z(rs_y, ok ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, o);
        }
    };
    if constexpr (MASK_) finish(std::false_type{}, std::true_type{});
    else if (p.relu != 0) finish(std::true_type{}, std::false_type{});
    else finish(std::false_type{}, std::false_type{});
    if (p.stamps && tid == 0) p.stamps[2 * blockIdx.x + 1] = wall_clock64();
}

template <class G, int DBG = 0, bool MASK_ = false>
int launch_win(WinArgs a, hipStream_t s) {
    constexpr int HW = G::H * G::W;
    a.n_mtiles = G::IMGS > 1 ? (a.N + G::IMGS - 1) / G::IMGS : a.N * G::TILES_PER_IMG;
    (void)HW;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_win_kernel<G, DBG, MASK_>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  G::LDS_BYTES);
        attr_set = true;
    }
    const int n_wg = a.n_mtiles * a.n_chunks;
    if (sc2_pol().win_stamps) {   // DEBUG: per-workgroup start / end times of this launch, summarised on stderr
        unsigned long long *d = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&d), (size_t)n_wg * 16) != hipSuccess) return SC2_ERR_INTERNAL;
        a.stamps = d;
        hipLaunchKernelGGL((conv3x3_win_kernel<G, DBG, MASK_>), dim3(n_wg), dim3(256), G::LDS_BYTES, s, a);
        (void)hipStreamSynchronize(s);
        unsigned long long *h = static_cast<unsigned long long *>(malloc((size_t)n_wg * 16));
        (void)hipMemcpy(h, d, (size_t)n_wg * 16, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < n_wg; ++i) { if (h[2 * i] < t0) t0 = h[2 * i]; if (h[2 * i + 1] > t1) t1 = h[2 * i + 1]; }
        double s_sum = 0, d_sum = 0, s_max = 0, d_max = 0, d_min = 1e30;
        for (int i = 0; i < n_wg; ++i) {
            const double so = (h[2 * i] - t0) * 0.01, du = (h[2 * i + 1] - h[2 * i]) * 0.01;   // us
            s_sum += so; d_sum += du; if (so > s_max) s_max = so; if (du > d_max) d_max = du; if (du < d_min) d_min = du;
        }
        fprintf(stderr, "conv3x3_win W=%d: %d workgroups, span %.2f us; start offset mean %.2f max %.2f us; duration min %.2f mean %.2f max %.2f us\n",
                G::W, n_wg, (t1 - t0) * 0.01, s_sum / n_wg, s_max, d_min, d_sum / n_wg, d_max);
        free(h);
        (void)hipFree(d);
        SC2_CHECK_LAUNCH();
        return SC2_OK;
    }
    hipLaunchKernelGGL((conv3x3_win_kernel<G, DBG, MASK_>), dim3(n_wg), dim3(256), G::LDS_BYTES, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

// Stride-2 form (conv2 of layer2.0 / layer3.0 / layer4.0: 56 -> 28, 28 -> 14, 14 -> 7): the same tile (196 output pixels x 128
// channels, wave w = 32 channels x 13 row tiles), the same k_step, the window in parity classes (GeoS2).  A stride-2 window
// holds 4x the pixels of its output tile (60 - 64 KB per 32-channel slab), so it is SINGLE-buffered and two workgroups share a
// CU: while one waits for its next window the other has the matrix pipes (a slab is 234 MFMAs per wave = 3.7 k cycles against
// ~2 k cycles to land 60 KB).  Two barriers per slab: everybody done with window cb - 1 / window cb complete.
template <class G>
__global__ __launch_bounds__(256, G::MT == 7 ? 3 : 2) void conv3x3s2_win_kernel(const WinArgs p) {
    constexpr int MT = G::MT, W = G::W, H = G::H, OW = G::OW, OH = G::OH;
    constexpr uint32_t OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cin = p.Cin, Cout = p.Cout;
    const int NS = Cin >> 5;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int chunk = bid % p.n_chunks, mtile = bid / p.n_chunks;
    const int n0 = chunk * 128 + wave * 32;
    const int img0 = G::IMGS > 1 ? mtile * G::IMGS : mtile / G::TILES_PER_IMG;
    const int row0 = G::IMGS > 1 ? 0 : (mtile % G::TILES_PER_IMG) * G::ROWS;   // first OUTPUT row of the tile

    const buf_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
    const buf_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

    // window fill: wave w fills chunk plane w; window row wr = (image, parity class, ihp >> 1, iwp >> 1)
    uint32_t pw_vo[G::NRG];
#pragma unroll
    for (int j = 0; j < G::NRG; ++j) {
        const int wr = j * 64 + lane;
        const int il = wr / G::IMGP, rem = wr - il * G::IMGP;
        const int cls = rem / G::CLS, rem2 = rem - cls * G::CLS;
        const int r = rem2 / G::PWD, c = rem2 - r * G::PWD;
        const int img = img0 + il, ih = 2 * (row0 + r) + (cls >> 1) - 1, iw = 2 * c + (cls & 1) - 1;
        const bool ok = (wr < G::WROWS) & (img < p.N) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
        pw_vo[j] = ok ? (uint32_t)((((img * H + ih) * W + iw) * Cin) * 2 + wave * 16) : OOB;
    }
    auto issue_window = [&](int cb) {
#pragma unroll
        for (int j = 0; j < G::NRG; ++j)
            buf_load_lds16(rs_x, (lds_ptr_t)(smem + wave * G::PLANE + j * 1024), pw_vo[j], (uint32_t)cb * 64u);
    };

    // fragment rows of this lane at tap (0, 0): class (0, 0), (ohl, ow)
    uint32_t a_base[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = i * 16 + frow;
        m = m < G::PX ? m : G::PX - 1;
        const int il = m / (G::ROWS * OW), rem = m - il * (G::ROWS * OW);
        const int ohl = rem / OW, ow = rem - ohl * OW;
        a_base[i] = lds_base + (uint32_t)(fq * G::PLANE + (il * G::IMGP + ohl * G::PWD + ow) * 16);
    }

    const uint32_t b_vo = (uint32_t)(lane * 16);
    const uint32_t b_step = (uint32_t)(Cout >> 4) * 1024u;
    const uint32_t KT = (uint32_t)NS * 9u;
    const uint32_t b_so0 = (uint32_t)(n0 >> 4) * 1024u;
    auto fetch_b = [&](uint32_t kt, uint4 &b0, uint4 &b1) {
        const uint32_t so = b_so0 + (kt < KT - 1u ? kt : KT - 1u) * b_step;
        b0 = buf_load16(rs_w, b_vo, so);
        b1 = buf_load16(rs_w, b_vo, so + 1024u);
    };

    f32x4_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    uint4 bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch_b((uint32_t)s, bq[s][0], bq[s][1]);

#define SC2_S2_STEP(cb, TAP, SLOT)                                                     \
    {                                                                                 \
        const uint4 b0 = bq[SLOT][0], b1 = bq[SLOT][1];                               \
        fetch_b((uint32_t)(cb) * 9u + (TAP + PF), bq[SLOT][0], bq[SLOT][1]);          \
        k_step<G, 0, TAP, 0>(acc, a_base, b0, b1);                                    \
    }
    for (int cb = 0; cb < NS; ++cb) {
        if (cb > 0) __builtin_amdgcn_s_barrier();   // everybody is done with window cb - 1
        issue_window(cb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // window cb complete
        SC2_S2_STEP(cb, 0, 0) SC2_S2_STEP(cb, 1, 1) SC2_S2_STEP(cb, 2, 2)
        SC2_S2_STEP(cb, 3, 0) SC2_S2_STEP(cb, 4, 1) SC2_S2_STEP(cb, 5, 2)
        SC2_S2_STEP(cb, 6, 0) SC2_S2_STEP(cb, 7, 1) SC2_S2_STEP(cb, 8, 2)
    }
#undef SC2_S2_STEP

    const float4 bias_lo = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq);
    const float4 bias_hi = *reinterpret_cast<const float4 *>(p.bias + n0 + 8 * fq + 4);
    const long long m_base = G::IMGS > 1 ? (long long)img0 * (OH * OW) : ((long long)img0 * OH + row0) * OW;
    const long long M = (long long)p.N * OH * OW;
    // Branch-free (round 4): the ReLU flag is tested once (as `if (relu)` inside the unrolled loop it was two scalar branches per
    // row tile) and masked lanes store out of range through a descriptor instead of jumping around the store -- a workgroup
    // lives for 3 - 6 us, and its ~60 branches were a measurable part of that.
    const buf_rsrc_t rs_y = make_rsrc(p.y, p.y_bytes);
    auto finish = [&](auto relu_c) {
        constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ml = i * 16 + frow;
            const long long m = m_base + ml;
            float v[8] = {acc[i][0][0] + bias_lo.x, acc[i][0][1] + bias_lo.y, acc[i][0][2] + bias_lo.z, acc[i][0][3] + bias_lo.w,
                          acc[i][1][0] + bias_hi.x, acc[i][1][1] + bias_hi.y, acc[i][1][2] + bias_hi.z, acc[i][1][3] + bias_hi.w};
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            const bool ok = (ml < G::PX) & (m < M);
            buf_store16_z(rs_y, ok ? (uint32_t)((m * Cout + n0 + 8 * fq) * 2) : 0x80000000u, o);
        }
    };
    if (p.relu != 0) finish(std::true_type{});
    else finish(std::false_type{});
}

template <class G>
int launch_win_s2(WinArgs a, hipStream_t s) {
    a.n_mtiles = G::IMGS > 1 ? (a.N + G::IMGS - 1) / G::IMGS : a.N * G::TILES_PER_IMG;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3s2_win_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  G::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3s2_win_kernel<G>), dim3(a.n_mtiles * a.n_chunks), dim3(256), G::LDS_BYTES, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

typedef Geo<28, 7, 1> G28;
typedef Geo<14, 14, 1> G14;
typedef Geo<7, 7, 4> G7;
// half tiles (7 row tiles, ~156 VGPRs, three workgroups per CU), the default: as fast alone (0.053 / 0.053 / 0.052 ms against
// 0.053 / 0.049 / 0.054), and launches of twice as many, smaller workgroups lose less when workgroup slots are taken by the
// kernels running beside them in the pipeline (bench + 0.5 % at K = 20, + 0.9 % at K = 100; SC2_WIN_HALF=0 = whole tiles;
// DESIGN.md section 6)
typedef Geo<28, 4, 1> H28;
typedef Geo<14, 7, 1> H14;
typedef Geo<7, 7, 2> H7;
typedef GeoS2<28, 7, 1> S28;   // 56 x 56 -> 28 x 28
typedef GeoS2<14, 14, 1> S14;  // 28 x 28 -> 14 x 14
typedef GeoS2<7, 7, 4> S7;     // 14 x 14 -> 7 x 7
typedef GeoS2<28, 4, 1> HS28;  // half tiles (SC2_WIN_HALF): 32 - 40 KB windows, three workgroups per CU
typedef GeoS2<14, 7, 1> HS14;
typedef GeoS2<7, 7, 2> HS7;

}  // namespace

extern "C" int sc2_conv3x3s2_win_supported(int H, int W, int Cin, int Cout) {
    if (H != W || (W != 56 && W != 28 && W != 14)) return 0;
    return Cin >= 32 && Cin % 32 == 0 && Cout >= 128 && Cout % 128 == 0 ? 1 : 0;
}

extern "C" int sc2_conv3x3s2_win_fwd(const void *x, const void *w_frag, const float *bias, void *y, int N, int H, int W, int Cin,
                                     int Cout, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv3x3s2_win: null argument");
    SC2_REQUIRE(N > 0, SC2_ERR_INVALID_ARG, "conv3x3s2_win: non-positive batch");
    SC2_REQUIRE(sc2_conv3x3s2_win_supported(H, W, Cin, Cout), SC2_ERR_UNSUPPORTED,
                "conv3x3s2_win: needs a 56 x 56, 28 x 28 or 14 x 14 input map, Cin %% 32 == 0, Cout %% 128 == 0 (got %d x %d, %d -> %d)", H,
                W, Cin, Cout);
    const long long x_bytes = (long long)N * H * W * Cin * 2, w_bytes = (long long)Cin * 9 * Cout * 2;
    const long long y_bytes = (long long)N * (H / 2) * (W / 2) * Cout * 2;
    SC2_REQUIRE(x_bytes < 0x7FF00000LL && w_bytes < 0x7FF00000LL && y_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED,
                "conv3x3s2_win: operand of %lld bytes exceeds 2 GB", x_bytes > w_bytes ? x_bytes : w_bytes);
    WinArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.mask = nullptr;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.relu = relu ? 1 : 0;
    a.n_chunks = Cout / 128; a.n_mtiles = 0;
    a.x_bytes = (unsigned)x_bytes; a.w_bytes = (unsigned)w_bytes; a.y_bytes = (unsigned)y_bytes;
    a.stamps = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int half = sc2_pol().win_half;
    if (half) {
        if (W == 56) return launch_win_s2<HS28>(a, s);
        if (W == 28) return launch_win_s2<HS14>(a, s);
        return launch_win_s2<HS7>(a, s);
    }
    if (W == 56) return launch_win_s2<S28>(a, s);
    if (W == 28) return launch_win_s2<S14>(a, s);
    return launch_win_s2<S7>(a, s);
}

extern "C" int sc2_conv3x3_win_supported(int H, int W, int Cin, int Cout) {
    if (H != W || (W != 28 && W != 14 && W != 7)) return 0;
    return Cin >= 64 && Cin % 64 == 0 && Cout >= 128 && Cout % 128 == 0 ? 1 : 0;
}

extern "C" int sc2_conv3x3_win_fwd(const void *x, const void *w_frag, const float *bias, const void *mask, void *y, int N, int H, int W,
                                   int Cin, int Cout, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv3x3_win: null argument");
    SC2_REQUIRE(!(mask && relu), SC2_ERR_INVALID_ARG, "conv3x3_win: mask and relu are exclusive");
    SC2_REQUIRE(N > 0, SC2_ERR_INVALID_ARG, "conv3x3_win: non-positive batch");
    SC2_REQUIRE(sc2_conv3x3_win_supported(H, W, Cin, Cout), SC2_ERR_UNSUPPORTED,
                "conv3x3_win: needs a 28 x 28, 14 x 14 or 7 x 7 map, Cin %% 64 == 0, Cout %% 128 == 0 (got %d x %d, %d -> %d)", H, W, Cin,
                Cout);
    const long long x_bytes = (long long)N * H * W * Cin * 2, w_bytes = (long long)Cin * 9 * Cout * 2;
    const long long y_bytes = (long long)N * H * W * Cout * 2;
    SC2_REQUIRE(x_bytes < 0x7FF00000LL && w_bytes < 0x7FF00000LL && y_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED,
                "conv3x3_win: operand of %lld bytes exceeds 2 GB", x_bytes > w_bytes ? x_bytes : w_bytes);
    WinArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.mask = static_cast<const uint16_t *>(mask);
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.relu = relu ? 1 : 0;
    a.n_chunks = Cout / 128; a.n_mtiles = 0;
    a.x_bytes = (unsigned)x_bytes; a.w_bytes = (unsigned)w_bytes; a.y_bytes = (unsigned)y_bytes;
    a.stamps = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int half = sc2_pol().win_half;
    if (a.mask) {   // the ReLU-gradient epilogue (training: data gradient of a frozen block's conv2)
        if (half) {
            if (W == 28) return launch_win<H28, 0, true>(a, s);
            if (W == 14) return launch_win<H14, 0, true>(a, s);
            return launch_win<H7, 0, true>(a, s);
        }
        if (W == 28) return launch_win<G28, 0, true>(a, s);
        if (W == 14) return launch_win<G14, 0, true>(a, s);
        return launch_win<G7, 0, true>(a, s);
    }
    if (half) {
        if (W == 28) return launch_win<H28>(a, s);
        if (W == 14) return launch_win<H14>(a, s);
        return launch_win<H7>(a, s);
    }
    if (W == 28) return launch_win<G28>(a, s);
    if (W == 14) {
#ifdef SC2_EXPERIMENTS   // timing experiments on the 14 x 14 geometry (results garbage): never in the shipped library
        switch (sc2_pol().win_dbg) {
            case 1: return launch_win<G14, 1>(a, s);
            case 2: return launch_win<G14, 2>(a, s);
            case 4: return launch_win<G14, 4>(a, s);
            case 8: return launch_win<G14, 8>(a, s);
            case 3: return launch_win<G14, 3>(a, s);
            case 7: return launch_win<G14, 7>(a, s);
            case 15: return launch_win<G14, 15>(a, s);
            default: break;
        }
#else
        SC2_REQUIRE(sc2_pol().win_dbg == 0, SC2_ERR_UNSUPPORTED, "conv3x3_win: win_dbg timing experiments need a -DSC2_EXPERIMENTS build");
#endif
        return launch_win<G14>(a, s);
    }
    return launch_win<G7>(a, s);
}
