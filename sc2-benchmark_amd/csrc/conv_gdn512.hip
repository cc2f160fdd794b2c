// First decoder stage of the FP bottleneck in ONE launch (gfx950):
//     y = IGDN1_512( Conv2d(Cin -> 512, k2, s1, p1, bias=False)(x) )          (sc2bench/models/layer.py:486-488)
// i.e.  t = conv(x);  y = t * (beta + gamma |t|)   (or t / (...) for the non-inverse GDN1).
//
// Why a dedicated kernel: as two launches the 512-channel intermediate t (822 MB at 256 x 56 x 56) is written once and
// read twice, and the K = 96 conv is all prologue/epilogue.  Here t never leaves the CU.
//
// Structure: persistent 512-thread workgroups (one per CU), each looping over tiles of 128 consecutive output pixels x
// all 512 channels.  Eight waves, wave w owns channels [64w, 64w+64) of every pixel of the tile (8 x 4 accumulator
// tiles of v_mfma_f32_16x16x32_bf16, weights as the A operand so a lane holds 4 consecutive channels of a pixel).
//   phase 1  t = W0 * patch:  the tile's im2col patch [128 px][4 taps x Cin] (<= 32 KB) sits in LDS (loaded one tile
//            ahead, during phase 2 of the previous tile); W0 fragments come straight from L2 into registers.
//            t is rounded to bf16 into the LDS IMAGE [128 px][512 ch] (128 KB, 16-byte chunks XOR-swizzled by row).
//   phase 2  norm = gamma * |t|:  K = 512; |t| fragments are read from the image (every wave reads all 128 rows),
//            gamma fragments stream from L2 into registers, two 16-byte halves of a 128-byte line per lane per
//            64-deep step (the k order inside a step is permuted identically for both operands so that the four
//            lanes of a row fetch one whole line).  No barrier inside the phase: the two waves of a SIMD drift
//            apart and one issues MFMAs while the other waits for its loads.
//   epilogue y = t * (beta + norm) in f32, written back IN PLACE into the image (each 8-byte slot belongs to one
//            lane), then the 128 KB tile - contiguous in the NHWC output - is streamed out in 16-byte stores.
// HBM traffic per tile: 128 x Cin x 2 B in (x taps overlap in L2), 128 KB out.
#include <stdlib.h>

#include <atomic>

#include "sc2_common.h"

#ifndef SC2_DEC_NT
#define SC2_DEC_NT 1       // non-temporal output stores (the 822 MB map is read by the NEXT launch, from HBM / the memory-side cache either way): - 2 %
#endif
#ifndef SC2_DEC_W_EARLY
#define SC2_DEC_W_EARLY 1
#endif
#ifndef SC2_DEC_EPI_BARRIER
#define SC2_DEC_EPI_BARRIER 1
#endif
#ifndef SC2_DEC_DBG
#define SC2_DEC_DBG 0      // timing experiments (results garbage): 1 = no output stores
#endif
#ifndef SC2_DEC_STAMPS
#define SC2_DEC_STAMPS 0   // 1: diagnostic build that records s_memtime at the phase boundaries (tools/dec_stamps.py)
#endif
#if SC2_DEC_STAMPS
#define STAMP(k)                                                                                      \
    do {                                                                                              \
        if (p.stamps && lane == 0 && blockIdx.x < 8 && n_done < 16)                                   \
            p.stamps[((blockIdx.x * 8 + wn) * 16 + n_done) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(k)
#endif

namespace {

// gamma fragments of phase 2 are loaded by INLINE ASM into a three-step register ring and waited for with hand-counted
// `s_waitcnt vmcnt(N)` (round 4).  Before, the ring was two steps deep on compiler-tracked loads: a step's fragments were
// requested one step (32 MFMAs) before their use and every step drained vmcnt to 0 (tools/audit_vmcnt.py) -- the L2 round trip
// was exposed whenever the partner wave of the SIMD was not exactly in antiphase (phase 2: 70 % MFMA-busy, tools/dec_stamps.py).
// The loads of step k + 3 go into the registers of step k behind that step's last MFMA (two whole steps = 64 MFMAs of this
// wave, ~2 k cycles with its partner, between request and use); vmcnt retires in issue order, so the wait for step k is "all
// but the 8 loads of the two steps behind it".
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32 on an explicit (e0, e1) pair.  With scalar element-wise code the SLP vectoriser paired the wrong
// elements ((e0, e2), (e1, e3)) and re-interleaved them with and / shift / or_sdwa / v_mov: 870 vector instructions per wave for
// the in-place epilogue of a tile where 350 do (round 4).
__device__ __forceinline__ uint32_t pack2(f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
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
// (the wait does not name the fragment registers -- as tied operands the compiler may copy them in front of the wait, i.e.
//  while their loads are in flight: conv2x2_win.hip, rule (ii).  The MFMAs of a step consume |t| fragments read from LDS by
//  ordinary loads, which the "memory" clobber keeps behind the wait.)
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct DecArgs {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N,H,W,CIN]
    const uint16_t *__restrict__ w;      // packed bf16 [512][Kpad], k = (kh*2+kw)*CIN + ci
    const uint16_t *__restrict__ g;      // bf16 gamma, fragment-major [32 channel tiles][16 k steps][64 lanes][8]
    const float *__restrict__ beta;      // f32 [512]
    uint16_t *__restrict__ y;            // bf16 NHWC [N,OH,OW,512]
    uint16_t *__restrict__ t_out;        // EMIT instantiations: the conv output t in front of the GDN (bf16, laid out like y)
    int N, H, W, OH, OW, OHW, M, Kpad, n_tiles, inverse;
    unsigned long long *stamps;          // diagnostic build only (SC2_DEC_STAMPS)
    unsigned *tile_ctr;                  // claims so far (claim c = tile c + 2 * gridDim.x); zero between launches
    int stagger;                         // start delay per workgroup of an XCD, in 64-cycle sleeps (see the launcher)
};

constexpr int BM = 128, CH = 512, WN = 64, MT = 8, NT = 4;
constexpr int IMG_BYTES = BM * CH * 2;

__device__ __forceinline__ int slab_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 3)) << 4); }
__device__ __forceinline__ int img_off(int row, int c16) { return row * (CH * 2) + ((c16 ^ (row & 15)) << 4); }

// EMIT (round 5, the training forward): t leaves too -- the tensor the GDN's backward needs, which this launch otherwise never
// materialises.  The image holds it, complete and read-only, for the whole of phase 2: every k-step sends ONE 16-byte chunk per thread
// (row wn + 8 ks of the tile, the mapping of the final read-out) behind its MFMAs, so that the sixteen stores of a tile are spread
// over the phase and have two steps each to retire in front of the gamma loads that are counted behind them (vmcnt retires in issue
// order, loads and stores alike).  The counted waits of the phase grow by the stores issued behind a step's own loads: 8, 9, 10, 10,
// ... 10, 6, 2 (tools/audit_vmcnt.py --counts follows them).  Rows past M (the last tile) are dropped by the descriptor's range
// check and keep their place in that order (tools/micro/oob_store_order.hip).
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;

template <int CIN, bool INVERSE, bool EMIT = false>
__global__ __launch_bounds__(512, 2) void conv2x2_gdn512_kernel(const DecArgs p) {
    constexpr int CIN8 = CIN / 8;
    constexpr int KS1 = (4 * CIN + 31) / 32;     // 32-deep k-steps of the conv (K = 4 taps x CIN)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *img = smem;
    unsigned char *patch = smem + IMG_BYTES;     // KS1 slabs of [128 rows][64 B]
    // (an LDS-space pointer: through a generic `volatile int *` the read was a flat_load followed by s_waitcnt vmcnt(0))
    typedef __attribute__((address_space(3))) volatile int lds_int_t;
    lds_int_t *next_slot = (lds_int_t *)(__attribute__((address_space(3))) unsigned char *)(smem + IMG_BYTES + KS1 * 8192);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int H = p.H, W = p.W;
    // LDS addresses as (per-lane base) + (compile-time constant), so that they fold into the ds instructions'
    // immediate offsets instead of occupying registers: row = i*16 + frow, so row & 15 = frow everywhere.
    // (ds offsets are 16 bits: bases above 64 KB are made opaque so that the constant part stays an immediate)
    int patch_lane = IMG_BYTES + frow * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);   // + s*8192 + i*1024
    asm volatile("" : "+v"(patch_lane));
    int slot_lane[NT];   // image slot of this lane's 4 channels of accumulator tile (i, j):  + i*16384
#pragma unroll
    for (int j = 0; j < NT; ++j)
        slot_lane[j] = frow * (CH * 2) + (((wn * 8 + j * 2 + (fq >> 1)) ^ frow) << 4) + (fq & 1) * 8;
    auto hi = [](int base) {   // base + 65536 as a value the compiler cannot fold back into a 17-bit offset
        int v = base + 65536;
        asm volatile("" : "+v"(v));
        return v;
    };

    // this thread's share of a tile's patch: chunk (row = tid >> 2, c = tid & 3) of each of the KS1 slabs
    const int prow = tid >> 2, pc_lane = tid & 3;
    auto load_patch = [&](int tile, uint4 (&pv)[KS1]) {
        int pc = pc_lane;   // (opaque per call: hoisted out of the tile loop, the per-slab source pointers derived from it were spilled)
        asm volatile("" : "+v"(pc));
        const int m = tile * BM + prow;
        const bool m_ok = (tile < p.n_tiles) & (m < p.M);
        const int mm = m_ok ? m : 0;
        const int im = mm / p.OHW;
        const int rem = mm - im * p.OHW;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const int kc = s * 4 + pc;               // 16-byte k chunk: (tap, 8-channel group)
            const int tap = kc / CIN8, c8 = kc - tap * CIN8;
            const int ih = oh - 1 + (tap >> 1), iw = ow - 1 + (tap & 1);
            const bool ok = m_ok & (tap < 4) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            const long long off = (((long long)im * H + ih) * W + iw) * CIN + c8 * 8;
            pv[s] = ok ? *reinterpret_cast<const uint4 *>(p.x + off) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_patch = [&](const uint4 (&pv)[KS1]) {
#pragma unroll
        for (int s = 0; s < KS1; ++s) *reinterpret_cast<uint4 *>(patch + s * 8192 + slab_off(prow, pc_lane)) = pv[s];
    };

    int tile = blockIdx.x;
    {
        uint4 pv[KS1];
        load_patch(tile, pv);
        store_patch(pv);
    }
    __syncthreads();


    // W0 fragments of this wave (its 64 channels x K): the same for every tile, but 48 registers are too many to
    // hold through phase 2, so they are re-fetched (L2) per tile - issued BEFORE the previous tile's output stores so
    // that waiting for them never waits for those stores (vmcnt retires in issue order).
    uint4 wv[KS1][NT];
    auto load_w = [&]() {
        // (row pointers rebuilt per call from an opaque copy of the lane's fragment coordinates: eight registers of pointers
        //  held across phase 2 pushed loop invariants of the patch loader into scratch once the gamma ring grew)
        int fr = frow, fk = fq;
        asm volatile("" : "+v"(fr), "+v"(fk));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint16_t *wrow = p.w + (long long)(wn * WN + j * 16 + fr) * p.Kpad + fk * 8;
#pragma unroll
            for (int s = 0; s < KS1; ++s) wv[s][j] = *reinterpret_cast<const uint4 *>(wrow + s * 32);
        }
    };
    load_w();
    // staggered start: workgroup j of its XCD begins j * stagger * 64 cycles late (launcher: why)
    for (int n = (int)(blockIdx.x >> 3) * p.stagger; n > 0; --n) __builtin_amdgcn_s_sleep(1);

    // Tiles are claimed dynamically, one atomic per tile: a workgroup whose CU is shared with other kernels (the range
    // coder's serial waves), or that starts late because its CU's LDS was taken, simply processes fewer tiles.
    // (claimed one tile ahead of use, so that the next patch can be fetched without waiting for the claim)
    int next_tile = tile + gridDim.x;
    int n_done = 0;
    (void)n_done;
    while (tile < p.n_tiles) {
        const int m0 = tile * BM;
        STAMP(0);
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        int slot_hi[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) slot_hi[j] = hi(slot_lane[j]);

        // ---------------------------------------------------------------- phase 1: t = conv(x)
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            bf16x8_t wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = __builtin_bit_cast(bf16x8_t, wv[s][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t xf = __builtin_bit_cast(
                    bf16x8_t, *reinterpret_cast<const uint4 *>(smem + patch_lane + s * 8192 + i * 1024));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[i][j], 0, 0, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's read-ahead (registers)
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                uint2 h;
                h.x = pack2(f32x2_t{acc[i][j][0], acc[i][j][1]});
                h.y = pack2(f32x2_t{acc[i][j][2], acc[i][j][3]});
                *reinterpret_cast<uint2 *>(img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * 16384) = h;
                acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(1);
        __syncthreads();   // the image holds all 512 channels of the tile; the patch has been consumed
        STAMP(2);

        // ---------------------------------------------------------------- phase 2: norm = gamma |t|
        // gamma fragments come FRAGMENT-MAJOR from memory ([16-channel tile][32-deep step][lane][8 k], packed on the
        // host): one wave-instruction reads 1 KB contiguous = 8 whole 128-byte lines.  (Row-major gamma made every
        // such load touch 16 lines for 64 bytes each, and the phase ran at the vector L1's request rate: 2.7k cycles
        // per 32-deep step against 1k of MFMA issue - tools/dec_stamps.py.)  gb[h] holds step 2d + h and is
        // re-filled for step 2d + 2 + h as soon as its MFMAs have been issued.
        constexpr int NS = CH / 32, GR = 3;   // k-steps; depth of the gamma ring (four steps deep the kernel spilled)
        // fragment (channel tile wn * 4 + j, step ks) of this lane: 16 bytes at ((wn * 4 + j) * NS + ks) * 1024 + lane * 16
        const i32x4_t rs_g = rsrc_words(p.g, (uint32_t)(CH / 16) * NS * 1024u);
        const uint32_t g_vo = (uint32_t)(lane * 16);
        const uint32_t g_so0 = (uint32_t)(wn * NT) * NS * 1024u;
        u32x4_t gb[GR][NT];
        auto fetch_g = [&](int ks, u32x4_t (&g)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; ++j) wload16(g[j], rs_g, g_vo, g_so0 + (uint32_t)(j * NS + ks) * 1024u);
        };
#pragma unroll
        for (int h = 0; h < GR; ++h) fetch_g(h, gb[h]);
        // EMIT: chunk `lane` of row wn + 8 r of the tile's t, r = 0 .. 15 (one per k-step)
        [[maybe_unused]] buf_rsrc_t rs_t;
        if constexpr (EMIT) rs_t = __builtin_amdgcn_make_buffer_rsrc(p.t_out, 0, (int)((uint32_t)p.M * (uint32_t)(CH * 2)), 0x00020000);
        [[maybe_unused]] const uint32_t t_so = (uint32_t)m0 * (uint32_t)(CH * 2);
        auto emit_t = [&](int r) {
            if constexpr (EMIT) {
                int l = lane;
                asm volatile("" : "+v"(l));   // (per call: nothing of this held across the steps)
                const int row15 = wn + 8 * (r & 1);
                const int a = row15 * (CH * 2) + ((l ^ row15) << 4) + (r >> 1) * 16384;
                const uint4 v = *reinterpret_cast<const uint4 *>(img + a);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, rs_t, wn * (CH * 2) + l * 16 + r * (8 * CH * 2), (int)t_so,
                                                       SC2_DEC_NT ? SC2_BUF_AUX_NT : 0);
            }
        };
        // one k-step: YOUNG = loads issued behind this step's own four (the steps still in flight behind it)
#define SC2_DEC_STEP(ks, h, YOUNG)                                                                                           \
    {                                                                                                                        \
        int fq_s = fq;   /* (EMIT: opaque per step -- with seven peeled steps the per-step addresses, invariant across tiles, were  \
                            hoisted out of the tile loop and spilled) */                                                     \
        if constexpr (EMIT) asm volatile("" : "+v"(fq_s));                                                                   \
        const int kc = (ks) * 4 + fq_s;   /* this lane's 16-byte k chunk of the step */                                      \
        const int rd_lane = frow * (CH * 2) + ((kc ^ frow) << 4);   /* + i*16384 */                                          \
        const int rd_hi = hi(rd_lane);                                                                                       \
        wait_vm<YOUNG>();                                                                                                    \
        bf16x8_t gf[NT];                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) gf[j] = __builtin_bit_cast(bf16x8_t, gb[h][j]);                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                     \
            uint4 v = *reinterpret_cast<const uint4 *>(img + (i < 4 ? rd_lane : rd_hi) + (i & 3) * 16384);                   \
            v.x &= 0x7FFF7FFFu; v.y &= 0x7FFF7FFFu; v.z &= 0x7FFF7FFFu; v.w &= 0x7FFF7FFFu;   /* |t| */                      \
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, v);                                                             \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[j], xf, acc[i][j], 0, 0, 0);                          \
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                                    \
    }
        static_assert(NS == 16 && GR == 3, "four trips of three steps, then steps 12 .. 15");
        // (EMIT: E = 1 -- step k's own loads went out behind step k - 3; behind them: store k - 2, the loads of step k + 1, store
        //  k - 1, the loads of step k + 2.  The first trip is peeled: steps 0 and 1 have no / one store behind their loads.)
        constexpr int E = EMIT ? 1 : 0;
        if constexpr (EMIT) {
            if ((wn >> 2) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            SC2_DEC_STEP(0, 0, 2 * NT) emit_t(0); fetch_g(GR + 0, gb[0]);
            SC2_DEC_STEP(1, 1, 2 * NT + 1) emit_t(1); fetch_g(GR + 1, gb[1]);
            SC2_DEC_STEP(2, 2, 2 * NT + 2) emit_t(2); fetch_g(GR + 2, gb[2]);
        }
#pragma unroll 1
        for (int d = E; d < 4; ++d) {
            // the two waves of a SIMD (w and w + 4) take turns at priority: with a fixed priority (or none: age decides)
            // one of them runs the phase at full speed and the other finishes what is left alone, at half speed
            if ((d ^ (wn >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            SC2_DEC_STEP(GR * d + 0, 0, 2 * NT + 2 * E) emit_t(GR * d + 0); fetch_g(GR * d + GR + 0, gb[0]);
            SC2_DEC_STEP(GR * d + 1, 1, 2 * NT + 2 * E) emit_t(GR * d + 1); fetch_g(GR * d + GR + 1, gb[1]);
            SC2_DEC_STEP(GR * d + 2, 2, 2 * NT + 2 * E) emit_t(GR * d + 2); fetch_g(GR * d + GR + 2, gb[2]);
        }
        {   // steps 12 .. 15: only step 12 still fetches (step 15)
            if ((wn >> 2) & 1) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
            SC2_DEC_STEP(12, 0, 2 * NT + 2 * E) emit_t(12); fetch_g(15, gb[0]);
            SC2_DEC_STEP(13, 1, 2 * NT + 2 * E) emit_t(13);
            SC2_DEC_STEP(14, 2, 1 * NT + 2 * E) emit_t(14);
            SC2_DEC_STEP(15, 0, 2 * E) emit_t(15);
        }
#undef SC2_DEC_STEP
        __builtin_amdgcn_s_setprio(0);
        STAMP(3);
        // Loads issued here, in this order (vmcnt retires in issue order): beta for the epilogue below, then this
        // thread's share of the NEXT tile's patch (consumed only after the output stores), then the claim of the tile
        // after next (lane 0 of wave 0; every wave learns it behind the next barrier).
        float4 b4v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b4v[j] = *reinterpret_cast<const float4 *>(p.beta + wn * WN + j * 16 + fq * 4);
        uint4 pv[KS1];
        load_patch(next_tile, pv);
        if (tid == 0) {
            const unsigned c = atomicAdd(p.tile_ctr, 1u);
            *next_slot = (int)(c + 2 * gridDim.x);
            if (c == (unsigned)(p.n_tiles - 1)) *p.tile_ctr = 0u;   // the launch's last claim re-arms the counter
        }

        // (round 5: every wave has read its last |t| fragment before any wave rewrites its slots.  The phase has no barrier inside and
        //  the in-place epilogue below used to start as soon as a wave's own last MFMA was issued: a wave more than ~2 k cycles ahead of
        //  the slowest one would have replaced t by y under that wave's last k-steps.  Never observed -- the beta loads above sit in
        //  between -- but nothing ruled it out.  SC2_DEC_EPI_BARRIER=0: A/B.)
        if (SC2_DEC_EPI_BARRIER) __builtin_amdgcn_s_barrier();
        // ---------------------------------------------------------------- epilogue: y = t * (beta + norm), in place
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x2_t b01 = {b4v[j].x, b4v[j].y}, b23 = {b4v[j].z, b4v[j].w};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned char *slot = img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * 16384;
                const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                // explicit (e0, e1) / (e2, e3) pairs: packed add / multiply / convert, no shuffles
                const f32x2_t t01 = {__builtin_bit_cast(float, xr.x << 16), __builtin_bit_cast(float, xr.x & 0xFFFF0000u)};
                const f32x2_t t23 = {__builtin_bit_cast(float, xr.y << 16), __builtin_bit_cast(float, xr.y & 0xFFFF0000u)};
                const f32x2_t n01 = b01 + f32x2_t{acc[i][j][0], acc[i][j][1]};
                const f32x2_t n23 = b23 + f32x2_t{acc[i][j][2], acc[i][j][3]};
                uint2 o;
                if (INVERSE) {
                    o.x = pack2(t01 * n01);
                    o.y = pack2(t23 * n23);
                } else {
                    o.x = pack2(t01 * f32x2_t{1.0f / n01[0], 1.0f / n01[1]});
                    o.y = pack2(t23 * f32x2_t{1.0f / n23[0], 1.0f / n23[1]});
                }
                *reinterpret_cast<uint2 *>(slot) = o;
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#if SC2_DEC_W_EARLY
            // the next tile's W0 fragments, fetched HALFWAY through the epilogue (half of the norm accumulators are dead: no
            // register pressure) instead of directly in front of the read-out, whose sixteen stores per wave then queue behind
            // twelve scattered loads: 0.527 -> 0.514 ms (round 4)
            if (j == 1) load_w();
#endif
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
        const int tile_after_next = __builtin_amdgcn_readfirstlane(*next_slot);
#if !SC2_DEC_W_EARLY
        load_w();          // next tile's W0 fragments, issued ahead of the output stores
#endif
        {
            // thread (wave wn, lane) streams chunk `lane` of rows wn, wn + 8, ...: row & 15 = wn or wn + 8
            uint4 *yo = reinterpret_cast<uint4 *>(p.y + (long long)m0 * CH) + tid;   // the tile is contiguous in y
            int rd_base[2];   // rows wn + 16k and wn + 8 + 16k
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                rd_base[par] = (wn + 8 * par) * (CH * 2) + ((lane ^ (wn + 8 * par)) << 4);
                asm volatile("" : "+v"(rd_base[par]));
            }
            if (m0 + BM <= p.M) {
                // whole tile (every tile but possibly the last): eight reads in flight, then their eight stores.  Row by row
                // -- read, wait, store, branch on the row bound -- the copy was a chain of sixteen LDS latencies per wave:
                // 9.2 k of a tile's 46.7 k cycles (tools/dec_stamps.py) for 16 KB per wave
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (inline asm: left to itself hipcc keeps two reads in flight and waits between the stores)
                    u32x4_t v[8];
                    const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(img + rd_base[0] + h * 65536);
                    const uint32_t l1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(img + rd_base[1] + h * 65536);
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(v[r])
                                     : "v"((r & 1) ? l1 : l0), "n"((r >> 1) * 16384));
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                                 :
                                 : "memory");
#if !(SC2_DEC_DBG & 1)   // (timing experiment: no output stores)
#pragma unroll
#if SC2_DEC_NT
                    for (int r = 0; r < 8; ++r) __builtin_nontemporal_store(v[r], reinterpret_cast<u32x4_t *>(yo + (8 * h + r) * 512));
#else
                    for (int r = 0; r < 8; ++r) yo[(8 * h + r) * 512] = __builtin_bit_cast(uint4, v[r]);
#endif
#else
                    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 1
                for (int r = 0; r < BM / 8; ++r) {
                    const int row = wn + r * 8;
                    if (m0 + row < p.M)   // wave-uniform
                        yo[r * 512] = *reinterpret_cast<const uint4 *>(img + rd_base[r & 1] + (r >> 1) * 16384);
                }
            }
        }
        store_patch(pv);   // (the patch region was last read in phase 1)
        STAMP(6);
        __syncthreads();   // image free for the next tile's phase 1; next patch visible to every wave
        STAMP(7);
        ++n_done;
        tile = next_tile;
        next_tile = tile_after_next;
    }
}

constexpr int kCtrRing = 256;
sc2_counter_ring g_ctr_ring;
std::atomic<unsigned> g_ctr_seq{0};

template <int CIN, bool INVERSE, bool EMIT>
int launch_dec(const DecArgs &a, hipStream_t s) {
    constexpr int KS1 = (4 * CIN + 31) / 32;
    constexpr int lds = IMG_BYTES + KS1 * 8192 + 16;   // + the next-tile slot
    static_assert(lds <= 160 * 1024, "image + patch must fit the CU's LDS");
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2x2_gdn512_kernel<CIN, INVERSE, EMIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int g_num_cus = sc2_device_cus();
    const int grid = a.n_tiles < g_num_cus ? a.n_tiles : g_num_cus;
    // tile counter: one of a ring of device words (launches in flight on different streams must not share one),
    // preset on the stream to the first unclaimed tile
    unsigned *slot = g_ctr_ring.launch_slot(s, kCtrRing, 1, g_ctr_seq);
    if (!slot) return SC2_ERR_INTERNAL;
    DecArgs b = a;
    b.tile_ctr = slot;
    {
        b.stagger = sc2_pol().dec_stagger;
    }
    b.stamps = nullptr;
#if SC2_DEC_STAMPS
    const char *stamp_path = getenv("SC2_DEC_STAMPS");
    const size_t stamp_bytes = 8 * 8 * 16 * 8 * sizeof(unsigned long long);
    if (stamp_path) {
        void *sp = nullptr;
        (void)hipMalloc(&sp, stamp_bytes);
        (void)hipMemsetAsync(sp, 0, stamp_bytes, s);
        b.stamps = static_cast<unsigned long long *>(sp);
    }
#endif
    hipLaunchKernelGGL((conv2x2_gdn512_kernel<CIN, INVERSE, EMIT>), dim3(grid), dim3(512), lds, s, b);
#if SC2_DEC_STAMPS
    if (b.stamps) {
        static unsigned long long host[8 * 8 * 16 * 8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host, b.stamps, stamp_bytes, hipMemcpyDeviceToHost);
        (void)hipFree(b.stamps);
        if (FILE *f = fopen(stamp_path, "wb")) {
            fwrite(host, 1, stamp_bytes, f);
            fclose(f);
        }
    }
#endif
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv2x2_gdn512_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return (Cin == 8 || Cin == 16 || Cin == 24) && Cout == 512 && KH == 2 && KW == 2 && stride == 1 &&
                   pad == 1 ? 1 : 0;
}

extern "C" int sc2_conv2x2_gdn512_fwd(const void *x, const void *w_packed, int Kpad, const void *gamma_packed,
                                      const float *beta, void *y, void *t_out, int N, int H, int W, int Cin, int inverse,
                                      void *stream) {
    SC2_REQUIRE(x && w_packed && gamma_packed && beta && y, SC2_ERR_INVALID_ARG, "conv2x2_gdn512: null argument");
    SC2_REQUIRE(N > 0 && H > 0 && W > 0, SC2_ERR_INVALID_ARG, "conv2x2_gdn512: non-positive dimension");
    SC2_REQUIRE(sc2_conv2x2_gdn512_supported(Cin, 512, 2, 2, 1, 1), SC2_ERR_UNSUPPORTED,
                "conv2x2_gdn512: Cin %d not in {8,16,24}", Cin);
    SC2_REQUIRE(Kpad == sc2_conv_weight_pitch(4 * Cin), SC2_ERR_INVALID_ARG, "conv2x2_gdn512: Kpad %d != %d", Kpad,
                sc2_conv_weight_pitch(4 * Cin));
    const long long M = (long long)N * (H + 1) * (W + 1);
    SC2_REQUIRE(M < 0x7FFFFFFFLL - 256, SC2_ERR_UNSUPPORTED, "conv2x2_gdn512: N*OH*OW = %lld exceeds 2^31", M);
    DecArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_packed);
    a.g = static_cast<const uint16_t *>(gamma_packed);
    a.beta = beta;
    a.y = static_cast<uint16_t *>(y);
    a.t_out = static_cast<uint16_t *>(t_out);
    SC2_REQUIRE(!t_out || M * (CH * 2) < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv2x2_gdn512: t_out needs N*OH*OW*1024 B < 2^31 (got %lld px)", M);
    a.N = N; a.H = H; a.W = W; a.OH = H + 1; a.OW = W + 1; a.OHW = a.OH * a.OW; a.M = (int)M;
    a.Kpad = Kpad; a.n_tiles = (int)((M + BM - 1) / BM); a.inverse = inverse ? 1 : 0; a.tile_ctr = nullptr; a.stamps = nullptr; a.stagger = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (t_out) {   // training: y and the conv output in front of the GDN
        switch (Cin) {
            case 8: return inverse ? launch_dec<8, true, true>(a, s) : launch_dec<8, false, true>(a, s);
            case 16: return inverse ? launch_dec<16, true, true>(a, s) : launch_dec<16, false, true>(a, s);
            default: return inverse ? launch_dec<24, true, true>(a, s) : launch_dec<24, false, true>(a, s);
        }
    }
    switch (Cin) {
        case 8: return inverse ? launch_dec<8, true, false>(a, s) : launch_dec<8, false, false>(a, s);
        case 16: return inverse ? launch_dec<16, true, false>(a, s) : launch_dec<16, false, false>(a, s);
        default: return inverse ? launch_dec<24, true, false>(a, s) : launch_dec<24, false, false>(a, s);
    }
}
