// GDN1 / inverse GDN1 over 512 or 256 channels with the WHOLE channel row of a pixel tile resident in LDS (gfx950) -- the training-time
// counterparts of conv_gdn512.hip's second phase (96 channels: gdn96_strips.hip):
//
//   forward   y = x * (beta + gamma |x|)            (inverse)      |  x / (beta + gamma |x|)
//   backward  n  = beta + gamma |x|
//             dd = g * n,  dn = g * x               (inverse)      |  dd = g / n,  dn = -dd * x / n
//             dx = dd + sign(x) * (gamma^T dn)                      d_norm = dn is written out (d_gamma = dn^T |x|, d_beta = colsum dn)
//
// (sc2bench/models/layer.py:486-491: the 512- and 256-channel inverse GDN1 behind the first two decoder convs; their backward is reached
//  through loss.backward() in script/task/image_classification.py:79.)
//
// Why: as launches of the 128 x 128 tile kernel (sc2_gdn1_bwd_gemm) the two C x C GEMMs of the 512-channel backward run at 0.32 - 0.36
// PFLOP/s and move the 822 MB tensors nine times (1.30 + 1.16 ms at 256 x 56 x 56; the plain GEMM without any epilogue already takes
// 0.83 ms: K = 512 is sixteen slabs through a three-deep LDS ring per tile, re-read by four channel tiles).  Here a workgroup owns 128
// pixels x all channels: x lands in the LDS image once (direct-to-LDS, 128 / 64 KB), gamma streams L2 -> registers as MFMA fragments
// (hand-counted vmcnt ring, as conv_gdn512.hip), the element-wise halves run on the accumulators in place, and -- backward -- the second
// GEMM accumulates ON TOP of sign(x) * dd in the same accumulators (dx = sign(x) * (sign(x) dd + gamma^T dn)), so neither n nor dd ever
// exists in memory: four tensor passes (x, g in; dn, dx out).  Measured at bs 256 (tools/gdn_gemm_times.py): 512 channels forward
// 0.88 -> 0.60 ms, backward 2.45 -> 1.22 ms; 256 channels 0.35 -> 0.22 and 0.92 -> 0.45 ms.
//
// sign(0) = 0 (torch.abs's gradient): an element with x == 0 contributes dn = 0 and must come out as dx = dd alone.  Its accumulator
// starts at 0, a per-lane bit mask remembers it, its dd is parked in its own place in dx (one 8-byte store per slot, sent out of
// range unless the slot holds a zero) and fetched back behind the second GEMM.
//
// Eight waves; wave w owns channels [CH/8 w, CH/8 (w + 1)) of every pixel (8 x 4 / 8 x 2 accumulator tiles, weights as the MFMA A
// operand: a lane holds 4 consecutive channels of one pixel).  Persistent workgroups, one per CU, tiles dealt round-robin.
//
// Three things this kernel taught (round 5), each now checked where it can be:
//   * a register an asm load is filling must never be under spilling pressure -- the compiler spills it right behind the load, i.e.
//     stores what the register held BEFORE (tools/audit_inflight.py; audit_vmcnt.py --copies);
//   * a VMEM store of more than 64 bits written in asm needs its own wait states before a VALU instruction overwrites the data
//     registers (store16_nt; audit_vmcnt.py --stores);
//   * __builtin_bit_cast(uint32_t, vec[e]) on an ext_vector element LVALUE reads element 0, whatever e is (copy the element first).
#include <stdlib.h>

#include "sc2_common.h"

#ifndef SC2_ROWS_GR
#define SC2_ROWS_GR 3
#endif

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ uint32_t pack2(f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ i32x4_t rsrc_words(const void *base, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// ("s_nop 4": a VMEM instruction needs 5 wait states behind a VALU write of an SGPR it reads, and the hazard recognizer does not
//  look into inline asm -- conv_gdn512.hip)
__device__ __forceinline__ void wload16(u32x4_t &d, i32x4_t r, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen ; wfrag" : "=&v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void park8(u32x2_t d, i32x4_t r, uint32_t voff) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, 0 offen ; park" ::"v"(d), "v"(voff), "s"(r) : "memory");
}
// (trailing "s_nop 1": a VMEM store of more than 64 bits needs wait states before a VALU instruction overwrites its data registers --
//  they are read out over several cycles -- and the hazard recognizer does not see the store inside the asm: without them the
//  v_lshl_add that computed the NEXT store's offset into the first data register corrupted dword 0 of a quarter of the lanes)
__device__ __forceinline__ void store16_nt(u32x4_t d, i32x4_t r, uint32_t voff) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen nt ; out\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ void gload8(u32x2_t &d, i32x4_t r, uint32_t voff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen ; gop" : "=&v"(d) : "v"(voff), "s"(r) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// ... naming the eight registers of a g group: what reads them stays behind the wait
template <int N>
__device__ __forceinline__ void wait_vm_q(u32x2_t (&q)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                 : "n"(N)
                 : "memory");
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, lds_ptr_t dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, (int)voff, (int)soff, 0, 0);
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
#endif

struct RowsArgs {
    const uint16_t *__restrict__ x;      // bf16 [M, 512]
    const uint16_t *__restrict__ gy;     // bf16 [M, 512]            (backward)
    const uint16_t *__restrict__ g1;     // gamma, fragment-major [32 channel tiles][16 k-steps][64 lanes][8]  (hip.pack_weight_fragments)
    const uint16_t *__restrict__ g2;     // gamma^T, the same packing  (backward)
    const float *__restrict__ beta;      // f32 [512]
    uint16_t *__restrict__ out;          // forward: y; backward: dx
    uint16_t *__restrict__ dn;           // backward: d_norm
    float *__restrict__ d_beta;          // backward, CH == 256: column sums of d_norm, accumulated here (zeroed by the launcher); else null
    int M, n_tiles;
    unsigned bytes;                      // M * 1024
};

constexpr int BM = 128, MT = 8, GR = SC2_ROWS_GR;
constexpr uint32_t OOB = 0x80000000u;

// Geometry of a channel count (512: the first decoder GDN; 256: the second).  A tile is always 128 pixels x all channels, eight waves
// with CH / 8 channels each.
template <int CH_>
struct RowsGeo {
    static constexpr int CH = CH_;
    static constexpr int WN = CH / 8, NT = WN / 16, NS = CH / 32;    // channels / accumulator column tiles per wave; k-steps
    static constexpr int ROWB = CH * 2;                              // bytes per image row
    static constexpr int I16 = 16 * ROWB;                            // bytes per 16-row block (16 384 / 8 192)
    static constexpr bool SPLIT = 8 * I16 > 65536;                   // ds offsets are 16 bits: rows 64 .. 127 need a second base
    static constexpr int CPR = CH / 8;                               // 16-byte chunks per row (64 / 32)
    static constexpr int RPP = 64 / CPR;                             // rows per 1 KB wave-instruction (1 / 2)
    static constexpr int PIECES = BM / RPP / 8;                      // direct-to-LDS instructions per wave and tile (16 / 8)
    static constexpr int STORES = BM * ROWB / 16 / 512;              // 16-byte output stores per thread and tile (16 / 8)
    static constexpr int MW = MT * NT / 8;                           // 32-bit words of a per-lane element bit mask (4 / 2)
    static constexpr int IMG_BYTES = BM * ROWB;
    static constexpr int MASK_OFF = IMG_BYTES + CH * 4;              // backward: [512 threads][2 masks x MW words]
    static constexpr int LDS_BYTES = MASK_OFF + 512 * 2 * MW * 4;
    static_assert(NT >= 2 && NT % 2 == 0 && CPR >= 16 && 64 % CPR == 0 && NS <= 16, "256 or 512 channels");
    // vmcnt budget of k-step ks: the fragment loads issued behind its own NT -- the steps the ring holds beyond it -- plus, for the
    // entry fetches, whatever else was issued between them and the GEMM (y0)
    static constexpr int young(int ks, int y0) {
        return ks < GR ? (GR - 1) * NT + y0 : ((NS - 1 - ks) < (GR - 1) ? (NS - 1 - ks) : (GR - 1)) * NT;
    }
};

// MODE 0: forward; 1: backward
template <int CH_, int MODE, bool INVERSE>
__global__ __launch_bounds__(512, 2) void gdn512_rows_kernel(const RowsArgs p) {
    typedef RowsGeo<CH_> G;
    constexpr int CH = G::CH, WN = G::WN, NT = G::NT, NS = G::NS, ROWB = G::ROWB, I16 = G::I16, IMG_BYTES = G::IMG_BYTES, MASK_OFF = G::MASK_OFF,
                  MW = G::MW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *img = smem;                                              // [128 px][512 ch] bf16, 16-byte chunks XORed with row & 15
    float *beta_s = reinterpret_cast<float *>(smem + IMG_BYTES);            // [512]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    if (tid < CH) beta_s[tid] = p.beta[tid];

    // image slot of this lane's 4 channels of accumulator tile (i, j): + i * I16
    int slot_lane[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) slot_lane[j] = frow * ROWB + (((wn * (WN / 8) + j * 2 + (fq >> 1)) ^ frow) << 4) + (fq & 1) * 8;

    auto hi = [](int base) {   // base + 65536 as a value the compiler cannot fold back into a 17-bit offset (ds offsets are 16 bits)
        int v = base + 65536;
        asm volatile("" : "+v"(v));
        return v;
    };

    const buf_rsrc_t rs_x = make_rsrc(p.x, p.bytes);
    const i32x4_t rs_gy = rsrc_words(p.gy ? p.gy : p.x, p.bytes);
    const i32x4_t rs_out = rsrc_words(p.out, p.bytes);
    const i32x4_t rs_dn = rsrc_words(MODE == 1 ? p.dn : p.out, p.bytes);
    const i32x4_t rs_g1 = rsrc_words(p.g1, (uint32_t)(CH / 16) * NS * 1024u);
    const i32x4_t rs_g2 = rsrc_words(MODE == 1 ? p.g2 : p.g1, (uint32_t)(CH / 16) * NS * 1024u);
    // fragment (channel tile wn * NT + j, step ks) of this lane: 16 bytes at ((wn * NT + j) * NS + ks) * 1024 + lane * 16.  The tile
    // index goes into the VECTOR offset (rebuilt per fetch from an opaque lane index: one v_lshl_add), the step into the scalar one:
    // as 64 different scalar offsets per GEMM the fully unrolled steps' offsets were all formed up front and spilled to VGPR lanes
    const uint32_t g_so0 = (uint32_t)(wn * NT) * NS * 1024u;
    u32x4_t gb[GR][NT];
    auto fetch_g = [&](const i32x4_t &rs, int ks, u32x4_t (&g)[NT]) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t so = g_so0 + (uint32_t)ks * 1024u;
#pragma unroll
        for (int j = 0; j < NT; ++j) wload16(g[j], rs, (uint32_t)(ln * 16 + j * NS * 1024), so);
    };

    // One k-step of a GEMM over the image: YOUNG = VMEM operations issued behind this step's own four fragment loads that may
    // still be in flight (vmcnt retires in issue order).  ABS: the operand is |image|.
#define SC2_ROWS_STEP(ks, h, YOUNG, ABS)                                                                                     \
    {                                                                                                                        \
        int fq_s = fq, fr_s = frow;   /* opaque: the sixteen steps' read offsets are rebuilt per step, not hoisted and spilled */ \
        asm volatile("" : "+v"(fq_s), "+v"(fr_s));                                                                           \
        const int kc = (ks) * 4 + fq_s;                                                                                      \
        const int rd_lane = fr_s * ROWB + ((kc ^ fr_s) << 4);   /* + i * I16 */                                              \
        const int rd_hi = G::SPLIT ? hi(rd_lane) : rd_lane + 4 * I16;                                                        \
        wait_vm<YOUNG>();                                                                                                    \
        bf16x8_t gf[NT];                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) gf[j] = __builtin_bit_cast(bf16x8_t, gb[h][j]);                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                     \
            uint4 v = *reinterpret_cast<const uint4 *>(img + (i < 4 ? rd_lane : rd_hi) + (i & 3) * I16);                     \
            if (ABS) { v.x &= 0x7FFF7FFFu; v.y &= 0x7FFF7FFFu; v.z &= 0x7FFF7FFFu; v.w &= 0x7FFF7FFFu; }                     \
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, v);                                                             \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[j], xf, acc[i][j], 0, 0, 0);                          \
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   /* bound the scheduler's read-ahead (registers) */        \
        }                                                                                                                    \
    }
    // the sixteen steps of one GEMM; the ring holds steps 0 .. GR - 1 on entry and step ks + GR is fetched into the registers of
    // step ks behind its last MFMA.  Y0: operations younger than the entry fetches that were issued before the GEMM starts (they count
    // against the waits of steps 0 .. GR - 1 only: every later fragment is fetched behind them).
    // The two waves of a SIMD (w and w + 4) take turns at priority every three steps (conv_gdn512.hip).
#define SC2_ROWS_ONE(rs, ks, ABS, Y0)                                                                                        \
    if constexpr ((ks) < NS) {                                                                                               \
        if ((ks) % 3 == 0) {                                                                                                 \
            if ((((ks) / 3) ^ (wn >> 2)) & 1) __builtin_amdgcn_s_setprio(1);                                                 \
            else __builtin_amdgcn_s_setprio(0);                                                                              \
        }                                                                                                                    \
        SC2_ROWS_STEP(ks, (ks) % GR, G::young(ks, Y0), ABS)                                                                  \
        if constexpr ((ks) + GR < NS) fetch_g(rs, (ks) + GR, gb[(ks) % GR]);                                                 \
    }
#define SC2_ROWS_GEMM(rs, ABS, Y0)                                                                                           \
    {                                                                                                                        \
        static_assert(NS <= 16, "at most sixteen steps");                                                                    \
        SC2_ROWS_ONE(rs, 0, ABS, Y0) SC2_ROWS_ONE(rs, 1, ABS, Y0) SC2_ROWS_ONE(rs, 2, ABS, Y0) SC2_ROWS_ONE(rs, 3, ABS, Y0)    \
        SC2_ROWS_ONE(rs, 4, ABS, Y0) SC2_ROWS_ONE(rs, 5, ABS, Y0) SC2_ROWS_ONE(rs, 6, ABS, Y0) SC2_ROWS_ONE(rs, 7, ABS, Y0)    \
        SC2_ROWS_ONE(rs, 8, ABS, Y0) SC2_ROWS_ONE(rs, 9, ABS, Y0) SC2_ROWS_ONE(rs, 10, ABS, Y0) SC2_ROWS_ONE(rs, 11, ABS, Y0)  \
        SC2_ROWS_ONE(rs, 12, ABS, Y0) SC2_ROWS_ONE(rs, 13, ABS, Y0) SC2_ROWS_ONE(rs, 14, ABS, Y0) SC2_ROWS_ONE(rs, 15, ABS, Y0) \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
    }

    // the image -> global, ALWAYS G::STORES sixteen-byte stores per thread (GEMM 2's first waits count them as younger operations),
    // every one of them in range.  (A store dropped by the descriptor's range check DOES keep its place in vmcnt's return order --
    // tools/micro/oob_store_order.hip, 5e8 probes -- so masking by range would have been safe too; this form dates from before that
    // was measured.)  A wave-instruction covers 1 KB = RPP rows; thread (wave wn, lane) streams
    // chunk lane % CPR of rows RPP (wn + 8 r) + lane / CPR; a row past the end of the tensor stores row 0 of the tile again (same data,
    // same address).  Branch-free and four rows at a time: this pass runs with the accumulators AND the next GEMM's fragment ring
    // live -- with eight rows in flight and a separate loop for the last tile the compiler spilled ring registers right behind
    // their (asm) loads, i.e. before the data had arrived (tools/audit_inflight.py).
    auto stream_out = [&](const i32x4_t &rs_dst, int m0) {
        constexpr int CPR = G::CPR, RPP = G::RPP;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int sub = ln / CPR, ch = ln % CPR;                 // row inside the instruction's RPP rows, chunk of the row
        const int ra = RPP * wn + sub, rb = RPP * (wn + 8) + sub;   // rows of r = 0 and r = 1; r + 2: 16 RPP rows further
        const uint32_t lz = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(img + (ch << 4));   // row 0
        const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(img + ra * ROWB + ((ch ^ (ra & 15)) << 4));
        const uint32_t l1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(img + rb * ROWB + ((ch ^ (rb & 15)) << 4));
        const uint32_t gz = (uint32_t)m0 * (uint32_t)ROWB + (uint32_t)(ch * 16);              // row m0
        const uint32_t g0 = gz + (uint32_t)(ra * ROWB);                                        // row m0 + ra (+ 8 RPP r)
        const int rows_left = p.M - m0 - ra;                                                   // row ra + 8 RPP r exists iff 8 RPP r < rows_left
        static_assert((16 * RPP) % 16 == 0, "r + 2 keeps row & 15");
#pragma unroll
        for (int q = 0; q < G::STORES / 4; ++q) {
            u32x4_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * q + k;   // (r & 1) picks the base, (r >> 1) * 16 RPP rows further
                const uint32_t la = ((r & 1) ? l1 : l0) + (uint32_t)((r >> 1) * 16 * RPP * ROWB);
                const uint32_t a = 8 * RPP * r < rows_left ? la : lz;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * q + k;
                store16_nt(v[k], rs_dst, 8 * RPP * r < rows_left ? g0 + (uint32_t)(8 * RPP * r * ROWB) : gz);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // d_beta = column sums of d_norm: with two accumulator columns per wave (CH == 256) a lane keeps the running sums of its 2 x 4
    // channels over all of the workgroup's tiles; the 512-channel form has no registers for sixteen (sc2_colsum_bf16 runs behind it)
    constexpr bool BSUM = MODE == 1 && NT <= 2;
    [[maybe_unused]] float bsum[NT][4];
    if constexpr (BSUM) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[j][e] = 0.f;
    }
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        // ---------------------------------------------------------------- x tile -> image: row r = one 1 KB wave-instruction;
        // LDS chunk position l of row r holds logical chunk l ^ (r & 15)
        int ln = lane;              // (opaque per tile: per-lane offsets are rebuilt where they are used -- hoisted out of the tile loop
        asm volatile("" : "+v"(ln));   //  they were held in scratch and reloaded with vmcnt(0) waits inside the counted pipeline)
#pragma unroll
        for (int k = 0; k < G::PIECES; ++k) {
            // piece wn + 8 k = rows RPP (wn + 8 k) .. + RPP - 1: lane -> (row, chunk position); position c of row r holds logical chunk c ^ (r & 15)
            const int row = G::RPP * (wn + 8 * k) + ln / G::CPR;
            const int m = m0 + row;
            const uint32_t vo = m < p.M ? (uint32_t)m * (uint32_t)ROWB + (uint32_t)(((ln % G::CPR) ^ (row & 15)) << 4) : OOB;
            buf_load_lds16(rs_x, (lds_ptr_t)(smem + (wn + 8 * k) * 1024), vo, 0u);
        }
#pragma unroll
        for (int h = 0; h < GR; ++h) fetch_g(rs_g1, h, gb[h]);
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        wait_vm<GR * NT>();     // this wave's rows of x have landed (the twelve fragment loads are younger)
        __syncthreads();        // ... everybody's

        // ---------------------------------------------------------------- GEMM 1: acc = gamma |x|
        SC2_ROWS_GEMM(rs_g1, true, 0)
        __syncthreads();        // every wave has read its last |x| fragment: the image may be rewritten in place
        int slot_hi[NT];        // (built here, not at the top of the tile: four registers less across the GEMM)
#pragma unroll
        for (int j = 0; j < NT; ++j) slot_hi[j] = G::SPLIT ? hi(slot_lane[j]) : slot_lane[j] + 4 * I16;

        if (MODE == 0) {
            // ------------------------------------------------------------ forward epilogue: y = x * (beta + norm), in place
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 b4 = *reinterpret_cast<const float4 *>(beta_s + wn * WN + j * 16 + fq * 4);
                const f32x2_t b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    unsigned char *slot = img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * I16;
                    const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
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
            }
            __syncthreads();
            stream_out(rs_out, m0);
            __syncthreads();    // the image is free for the next tile's x
            continue;
        }

        // ---------------------------------------------------------------- backward, first half: dn -> image, sign(x) dd -> acc
        // VMEM order from here: g in four column groups of eight 8-byte loads, each group requested one group ahead of its use
        uint32_t zmask[MW] = {}, smask[MW] = {};   // bit (i * NT + j) * 4 + e: x == 0 / x < 0
        u32x2_t gq[2][MT];
        auto load_gy = [&](int j, u32x2_t (&q)[MT]) {
            int fr = frow, fk = fq;
            asm volatile("" : "+v"(fr), "+v"(fk));   // (offsets rebuilt per group, not held across the half)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + i * 16 + fr;
                gload8(q[i], rs_gy, m < p.M ? (uint32_t)m * (uint32_t)ROWB + (uint32_t)((wn * WN + j * 16 + fk * 4) * 2) : OOB);
            }
        };
        load_gy(0, gq[0]);
        load_gy(1, gq[1]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // issue order: L0 L1 | C0 (8 parking stores) L2 | C1 (8) L3 | C2 (8) | C3 (8) | F (12 gamma^T fragments).  The waits rely
            // only on the LOADS behind the group they wait for, i.e. they also retire the previous group's parking stores (which are
            // nearly all dropped by the range check and cost nothing; such stores do keep their place in the return order --
            // tools/micro/oob_store_order.hip -- so counting them would be correct as well)
            if (j < NT - 1) wait_vm_q<MT>(gq[j & 1]);             // group j has landed: group j + 1 is younger
            else wait_vm_q<0>(gq[j & 1]);
            const float4 b4 = *reinterpret_cast<const float4 *>(beta_s + wn * WN + j * 16 + fq * 4);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
            int fr_o = frow, fq_o = fq;
            asm volatile("" : "+v"(fr_o), "+v"(fq_o));
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned char *slot = img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * I16;
                const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                const uint32_t xb[4] = {xr.x << 16, xr.x & 0xFFFF0000u, xr.y << 16, xr.y & 0xFFFF0000u};
                const u32x2_t gr = gq[j & 1][i];
                const float gv[4] = {__builtin_bit_cast(float, gr[0] << 16), __builtin_bit_cast(float, gr[0] & 0xFFFF0000u),
                                     __builtin_bit_cast(float, gr[1] << 16), __builtin_bit_cast(float, gr[1] & 0xFFFF0000u)};
                float dnv[4], ddv[4], sdd[4];
                uint32_t zm4 = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xv = __builtin_bit_cast(float, xb[e]);
                    const float norm = b[e] + acc[i][j][e];
                    if (INVERSE) {
                        dnv[e] = gv[e] * xv;
                        ddv[e] = gv[e] * norm;
                    } else {
                        const float rn = 1.0f / norm;
                        ddv[e] = gv[e] * rn;
                        dnv[e] = -ddv[e] * xv * rn;
                    }
                    // integer masks, no compares (as conditions the 128 of them were kept alive in SGPR pairs, spilled lane by lane):
                    // nz = all ones unless x is +-0
                    const uint32_t nz = (uint32_t)((int32_t)(0u - (xb[e] & 0x7FFFFFFFu)) >> 31);
                    zm4 |= (~nz & 1u) << e;
                    const int bit = ((i * NT + j) & 7) * 4 + e;
                    smask[(i * NT + j) >> 3] |= (xb[e] >> 31) << bit;
                    // sign(x) * dd: dd with its sign flipped where x < 0; an element with x == 0 starts from 0 (its dd is parked, below)
                    sdd[e] = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, ddv[e]) ^ (xb[e] & 0x80000000u)) & nz);
                }
                // (pin the four values HERE: left alone, the compiler sinks these selects to their use in GEMM 2 and keeps dd, x and the
                //  masks of all 128 elements alive until then -- a kilobyte of scratch per lane.  Four scalar operands: with the
                //  f32x4 tile as ONE "+v" operand the compiler took all four elements to be element 0 afterwards)
                if constexpr (BSUM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[j][e] += dnv[e];
                }
                asm volatile("" : "+v"(sdd[0]), "+v"(sdd[1]), "+v"(sdd[2]), "+v"(sdd[3]));
                acc[i][j] = f32x4_t{sdd[0], sdd[1], sdd[2], sdd[3]};
                zmask[(i * NT + j) >> 3] |= zm4 << (((i * NT + j) & 7) * 4);
                // sign(0) = 0: such an element's gradient is dd alone.  The slot's four dd are parked in their own place in dx (which
                // the tile's final store pass overwrites -- with the same value where x == 0) and fetched back behind the second GEMM.
                // Branch-free: ALWAYS one 8-byte store per slot, sent out of range unless the slot holds a zero (the counted waits
                // below do not rely on these stores; rows past the end of the tensor are out of the descriptor's range by themselves).
                {
                    const uint32_t po = (uint32_t)(m0 + i * 16 + fr_o) * (uint32_t)ROWB + (uint32_t)((wn * WN + j * 16 + fq_o * 4) * 2);
                    park8(u32x2_t{pack2(f32x2_t{ddv[0], ddv[1]}), pack2(f32x2_t{ddv[2], ddv[3]})}, rs_out, zm4 != 0u ? po : OOB);
                }
                uint2 o;
                o.x = pack2(f32x2_t{dnv[0], dnv[1]});
                o.y = pack2(f32x2_t{dnv[2], dnv[3]});
                *reinterpret_cast<uint2 *>(slot) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + 2 < NT) load_gy(j + 2, gq[j & 1]);
        }
        // the first three steps of gamma^T.  NOT earlier: requested inside the half above (behind group 2), two of the twelve ring
        // registers were spilled by the compiler right behind their loads -- i.e. before the data had arrived -- and GEMM 2 ran on
        // what the registers held before (registers with an asm load in flight must never be under spilling pressure)
#pragma unroll
        for (int h = 0; h < GR; ++h) fetch_g(rs_g2, h, gb[h]);
        // the two masks sit out GEMM 2 in LDS (eight registers the GEMM does not have: with them live across it the accumulators
        // were spilled)
        {
            uint32_t *mk = reinterpret_cast<uint32_t *>(smem + MASK_OFF) + 2 * MW * tid;
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                mk[w] = zmask[w];
                mk[MW + w] = smask[w];
            }
        }
        __syncthreads();        // the image holds dn
        stream_out(rs_dn, m0);   // sixteen stores, younger than the three gamma^T steps in the ring

        // ---------------------------------------------------------------- GEMM 2 on top: acc = sign(x) dd + gamma^T dn
        SC2_ROWS_GEMM(rs_g2, false, G::STORES)   /* younger than the ring's three steps: the dn stores */
        __syncthreads();        // every wave has read its last dn fragment

        // ---------------------------------------------------------------- dx = sign(x) * acc, in place
        {
            const uint32_t *mk = reinterpret_cast<const uint32_t *>(smem + MASK_OFF) + 2 * MW * tid;
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                zmask[w] = mk[w];
                smask[w] = mk[MW + w];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) slot_hi[j] = G::SPLIT ? hi(slot_lane[j]) : slot_lane[j] + 4 * I16;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned char *slot = img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * I16;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = ((i * NT + j) & 7) * 4 + e;
                    const uint32_t sb = ((smask[(i * NT + j) >> 3] >> bit) & 1u) << 31;
                    const float a = acc[i][j][e];   // (a copy: __builtin_bit_cast of the vector-element lvalue itself reads element 0)
                    o[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) ^ sb);
                }
                uint2 w;
                w.x = pack2(f32x2_t{o[0], o[1]});
                w.y = pack2(f32x2_t{o[2], o[3]});
                *reinterpret_cast<uint2 *>(slot) = w;
            }
        }
        // elements with x == 0: dx = the dd parked in front of GEMM 2
        uint32_t zany = 0u;
#pragma unroll
        for (int w = 0; w < MW; ++w) zany |= zmask[w];
        if (__builtin_amdgcn_ballot_w64(zany != 0u) != 0ull) {
            wait_vm<0>();   // the parking stores have completed (vmcnt counts stores on gfx9)
            int fr_o = frow, fq_o = fq;
            asm volatile("" : "+v"(fr_o), "+v"(fq_o));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint32_t zm4 = (zmask[(i * NT + j) >> 3] >> (((i * NT + j) & 7) * 4)) & 15u;
                    if (zm4 != 0u && m0 + i * 16 + fr_o < p.M) {   // (rows past the end of the tensor read zeros: flagged, never stored)
                        const volatile uint16_t *park = p.out + (long long)(m0 + i * 16 + fr_o) * CH + wn * WN + j * 16 + fq_o * 4;
                        unsigned char *slot = img + (i < 4 ? slot_lane[j] : slot_hi[j]) + (i & 3) * I16;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((zm4 >> e) & 1u) *reinterpret_cast<uint16_t *>(slot + e * 2) = park[e];
                    }
                }
            }
        }
        __syncthreads();
        stream_out(rs_out, m0);
        __syncthreads();        // the image is free for the next tile's x
    }
    if constexpr (BSUM) {
        if (p.d_beta) {   // the sixteen pixel lanes of a channel quad, then one atomic per channel and wave
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = bsum[j][e];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                    if (frow == 0) atomicAdd(p.d_beta + wn * WN + j * 16 + fq * 4 + e, v);
                }
        }
    }
#undef SC2_ROWS_GEMM
#undef SC2_ROWS_ONE
#undef SC2_ROWS_STEP
}

template <int CH, int MODE, bool INVERSE>
int launch_rows(const RowsArgs &a, hipStream_t s) {
    constexpr int lds = RowsGeo<CH>::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gdn512_rows_kernel<CH, MODE, INVERSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds);
        attr_set = true;
    }
    const int g_cus_rows = sc2_device_cus();
    const int grid = a.n_tiles < g_cus_rows ? a.n_tiles : g_cus_rows;
    hipLaunchKernelGGL((gdn512_rows_kernel<CH, MODE, INVERSE>), dim3(grid), dim3(512), lds, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

template <int MODE>
int launch_rows_c(const RowsArgs &a, int C, int inverse, hipStream_t s) {
    if (C == 512) return inverse ? launch_rows<512, MODE, true>(a, s) : launch_rows<512, MODE, false>(a, s);
    return inverse ? launch_rows<256, MODE, true>(a, s) : launch_rows<256, MODE, false>(a, s);
}

int rows_args(RowsArgs &a, const void *x, const void *gy, const void *g1, const void *g2, const float *beta, void *out, void *dn,
              long long M, int C, const char *who) {
    a.d_beta = nullptr;
    SC2_REQUIRE(M > 0 && M * (C * 2) < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "%s: %lld pixels x %d channels exceed 2 GB (32-bit buffer offsets)",
                who, M, C);
    a.x = static_cast<const uint16_t *>(x);
    a.gy = static_cast<const uint16_t *>(gy);
    a.g1 = static_cast<const uint16_t *>(g1);
    a.g2 = static_cast<const uint16_t *>(g2);
    a.beta = beta;
    a.out = static_cast<uint16_t *>(out);
    a.dn = static_cast<uint16_t *>(dn);
    a.M = (int)M;
    a.n_tiles = (int)((M + BM - 1) / BM);
    a.bytes = (unsigned)(M * (C * 2));
    return SC2_OK;
}

}  // namespace

// C = 96: every wave an independent worker on 32-pixel strips (gdn96_strips.hip)
int sc2_gdn96_strips(int mode, const void *x, const void *gy, const void *g1, const void *g2, const float *beta, void *out, void *dn,
                     float *d_beta, long long M, int inverse, hipStream_t s);
extern "C" int sc2_colsum_bf16(const void *x, long long M, int C, float *out, void *stream);

extern "C" int sc2_gdn1_rows_supported(int C) { return C == 512 || C == 256 || C == 96 ? 1 : 0; }

extern "C" int sc2_gdn1_rows_fwd(const void *x, const void *gamma_frag, const float *beta, void *y, long long M, int C, int inverse,
                                 void *stream) {
    SC2_REQUIRE(x && gamma_frag && beta && y, SC2_ERR_INVALID_ARG, "gdn1_rows_fwd: null argument");
    SC2_REQUIRE(sc2_gdn1_rows_supported(C), SC2_ERR_UNSUPPORTED, "gdn1_rows_fwd: C = %d (96, 256 or 512)", C);
    if (C == 96) return sc2_gdn96_strips(0, x, nullptr, gamma_frag, nullptr, beta, y, nullptr, nullptr, M, inverse, static_cast<hipStream_t>(stream));
    RowsArgs a;
    if (const int rc = rows_args(a, x, nullptr, gamma_frag, nullptr, beta, y, nullptr, M, C, "gdn1_rows_fwd")) return rc;
    return launch_rows_c<0>(a, C, inverse, static_cast<hipStream_t>(stream));
}

extern "C" int sc2_gdn1_rows_bwd(const void *x, const void *gy, const void *gamma_frag, const void *gamma_t_frag, const float *beta,
                                 void *d_norm, void *dx, float *d_beta, long long M, int C, int inverse, void *stream) {
    SC2_REQUIRE(x && gy && gamma_frag && gamma_t_frag && beta && d_norm && dx, SC2_ERR_INVALID_ARG, "gdn1_rows_bwd: null argument");
    SC2_REQUIRE(sc2_gdn1_rows_supported(C), SC2_ERR_UNSUPPORTED, "gdn1_rows_bwd: C = %d (96, 256 or 512)", C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d_beta && C != 512) {   // accumulated by the kernels (atomics)
        hipError_t e = hipMemsetAsync(d_beta, 0, (size_t)C * sizeof(float), s);
        SC2_REQUIRE(e == hipSuccess, SC2_ERR_LAUNCH, "gdn1_rows_bwd: memset failed: %s", hipGetErrorString(e));
    }
    if (C == 96) return sc2_gdn96_strips(1, x, gy, gamma_frag, gamma_t_frag, beta, dx, d_norm, d_beta, M, inverse, s);
    RowsArgs a;
    if (const int rc = rows_args(a, x, gy, gamma_frag, gamma_t_frag, beta, dx, d_norm, M, C, "gdn1_rows_bwd")) return rc;
    a.d_beta = C == 256 ? d_beta : nullptr;
    if (const int rc = launch_rows_c<1>(a, C, inverse, s)) return rc;
    if (d_beta && C == 512) return sc2_colsum_bf16(d_norm, M, C, d_beta, stream);   // (no registers for the running sums there)
    return SC2_OK;
}
