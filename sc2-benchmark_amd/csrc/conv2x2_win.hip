// The two MFMA-bound decoder layers of the FP / SHP / MSHP bottlenecks on the window-plane structure (gfx950):
//     dec.conv2 (512 -> 256, k2, p0: 56 -> 55) + inverse GDN1(256)   (sc2bench/models/layer.py:489-491)
//     dec.conv4 (256 -> 256, k2, p1: 55 -> 56)                        (layer.py:492-493)
// Successor of conv_dec_persist.hip, whose K loop is bound by what it moves L2 -> LDS (a 256 x 256 tile stages 16 KB of
// gathered pixels + 16 KB of weights per 32-deep slab; measured: 0.72 ms, 0.52 with either operand's loads switched off,
// 0.40 with both).  Here (the structure of conv3x3_win.hip, which see):
//   * a tile = 4 output rows x all 55 / 56 columns (14 MFMA row tiles) x all 256 channels; 8 waves, wave w owns channels
//     [32 w, 32 w + 32) for every row tile (28 accumulator tiles);
//   * per 32-channel slab the tile's 5-row input window is staged ONCE, zero-padded, as four 16-byte-chunk planes
//     [chunk][window row][16 B]: the four taps read it at immediate offsets (kh PWD + kw) * 16 -- 4.5 KB per k-step through
//     the L2 -> LDS path instead of 32 KB;
//   * the weights never touch LDS: two buffer_load_dwordx4 per wave and k-step, four k-steps ahead, fragment-major;
//   * one barrier per slab (112 MFMAs per wave); the window of the next slab -- or of the NEXT TILE's first slab -- is in
//     flight during the current one, so the K loops of successive tiles join without a bubble (persistent workgroups,
//     one per CU, XCD-contiguous tile ranges);
//   * fused inverse GDN1: the tile's conv output goes to a bf16 image in LDS (32 channel-chunk planes of 224 rows),
//     norm = gamma |x| is a second GEMM over the image (8 k-steps, gamma fragments through the same register ring),
//     y = x * (beta + norm) with x read back from the lane's own image slots; 14 sixteen-byte stores per lane.
//     Same operation order per output element as the tile kernels: results are bit-identical to theirs.
#include <stdlib.h>

#include "sc2_common.h"

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
// A 16-byte buffer store reads its data registers for a few cycles after issue (lanes 12-15 of every 16 last): the GCN / CDNA
// hazard "VMEM store of more than 64 bits followed by a VALU write of its data VGPRs".  hipcc's hazard recogniser inserts the
// wait states only when the store's soffset field is NOT a register (GCNHazardRecognizer exempts MUBUF stores with an SGPR
// soffset) -- and every store here carries an SGPR soffset.  On gfx950 the hazard exists in that form too:
// tools/micro/store_hazard.hip issues `buffer_store_dwordx4 v[10:13], voff, rsrc, sN offen` directly followed by writes of
// v10..v13 and finds the NEW value in memory for 0.12 % of the elements, all of them in lanes 12-15 of a group of 16 (with a
// literal soffset, the case the compiler does handle: 6 %); ONE wait state removes every error (profiles/r04_store_hazard.txt).
// That is what round 2 saw in this kernel: hipcc scheduled a VALU write of a data register directly behind a store, and lanes
// 12-15 of the second wave of a SIMD (the one that is delayed at the memory pipe) stored the new value.  Two wait states follow
// every store here (what the compiler inserts for the literal-soffset form on gfx940+); the bit-exact multi-launch test
// (tests/test_gpu_kernels.py::test_conv2x2_win) stays in the default GPU suite as the guard against a compiler update.
__device__ __forceinline__ void buf_store16(buf_rsrc_t r, uint32_t voff, uint32_t soff, u32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t, uint32_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void buf_store16(buf_rsrc_t, uint32_t, uint32_t, u32x4_t) {}
#endif

// Weight fragments are loaded by INLINE ASM and waited for with hand-counted `s_waitcnt vmcnt(N)` (round 4).  With the
// compiler's own loads the conditional window fills in front of a slab made its scoreboard merge conservative: the first
// k-step of every other slab waited `vmcnt(2)`, i.e. for the window pieces of the NEXT slab issued a few instructions
// earlier (a full L2 / HBM round trip with every wave of the workgroup parked; tools/audit_vmcnt.py shows such waits).
// The asm loads are invisible to that scoreboard; vmcnt retires in issue order, so "all but the N youngest" is exact.
// Rules that keep this safe: (i) a fragment register is written by its load and read only by the MFMAs of its k-step,
// which sit behind wait_vm: every MFMA of a step consumes a pixel fragment that went through wait_lgkm (asm volatile, issued
// behind wait_vm in program order), so none can be placed above it; (ii) wait_vm does NOT name the registers: as a tied
// "+v" operand the compiler once placed a v_mov of the fragment into a fresh register IN FRONT of the wait -- a copy of a
// register whose load is still in flight (stale weights, found by the bit-exact multi-tile test); (iii) the load of k-step
// k + 4 goes into the registers of k-step k AFTER that step's last MFMA (no renaming); (iv) tools/audit_vmcnt.py --copies
// checks in the built ISA that no instruction but an MFMA reads a register written by such a load (tests/test_abi.py).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t rsrc_words(const void *base, uint32_t bytes) {   // raw buffer descriptor: base, no stride, size, 32-bit raw data format
    const uint64_t a = (uint64_t)(uintptr_t)base;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// ("s_nop 4": the hazard recognizer does not look into inline asm.  The scalar offset / descriptor may have been written by a
//  VALU instruction just before -- hipcc restores spilled SGPRs with v_readlane_b32 -- and a VMEM instruction needs 5 wait states
//  behind a VALU write of an SGPR it reads: without them the loads of the tail-mode kernel used a stale offset (wrong weights) and
//  conv1x1_win a stale descriptor (memory fault).)
__device__ __forceinline__ void wload16(u32x4_t &d, i32x4_t r, uint32_t voff, uint32_t soff) {   // ("; wfrag": marker for the audit)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen ; wfrag" : "=&v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void cload16(u32x4_t &d, i32x4_t r, uint32_t voff, uint32_t soff) {   // epilogue constants (wait_vm_tied)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
#ifndef SC2_W2_VMCAP
#define SC2_W2_VMCAP 63   // DEBUG: -DSC2_W2_VMCAP=<n> caps every counted weight wait at n (0 = drain): bisects a wrong budget
#endif
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < SC2_W2_VMCAP ? N : SC2_W2_VMCAP) : "memory");
}
// (epilogue constants only: consumed by vector ALU code, which nothing else orders behind the wait.  Their loads are at least
//  eight k-steps old at the wait, so even a copy placed in front of it reads landed data)
template <int N>
__device__ __forceinline__ void wait_vm_tied(u32x4_t &a, u32x4_t &b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ u32x4_t lds_read16_imm(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write16_imm(uint32_t addr, u32x4_t v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// wait until at most N of this wave's LDS operations are outstanding; `v` (the destination of the read being waited for)
// is threaded through so that its consumers cannot be scheduled above the wait
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4_t &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

struct W2Args {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, W, Cin]
    const uint16_t *__restrict__ w;      // bf16 [Cin/32 * 4 (+ 8)][16][64][8]: conv k-steps, then (fused) the 8 gamma k-steps
    const float *__restrict__ beta;      // f32 [256] (fused) or null
    uint16_t *__restrict__ y;            // bf16 NHWC [N, OH, OW, 256]  (tail mode: may be null)
    // tail mode (MODE 2): the two 1x1 layers that consume y in the caller (layer2.0 of the ResNet tail) fused behind the conv
    const float *__restrict__ bias1;     // f32 [128]
    const float *__restrict__ bias_ds;   // f32 [512]
    uint16_t *__restrict__ o1;           // bf16 NHWC [N, OH, OW, 128]     = relu(W1 y + bias1)
    uint16_t *__restrict__ ods;          // bf16 NHWC [N, OH/2, OW/2, 512] = Wds y[::2, ::2] + bias_ds
    int N, H, Cin, OH;
    int W, OW, nseg;                     // runtime-geometry instantiations only (Geo2<.., .., true>): map widths, column segments per row block
    int n_tiles, tiles_per_img, tiles_per_wg;
    unsigned x_bytes, w_bytes, y_bytes, o1_bytes, ods_bytes;
};

// OW: output width (static: the tap offsets are immediates); PAD: 0 (W = OW + 1) or 1 (W = OW - 1).
// Round 4: the window row pitch AND the pitch of the tile's pixel index are both 56, so tile pixel m reads window row
// m + kh 56 + kw: sixteen consecutive fragment rows are sixteen consecutive 16-byte LDS rows for every row tile and tap --
// conflict-free under the ds_read_b128 lane groups (with a 55-pixel pitch the three row tiles that straddle an output row
// skipped a window row, rows R and R + 16 met in one group: 2.9e7 SQ_LDS_BANK_CONFLICT cycles per launch, profiles/r03f),
// and the fourteen fragment addresses are ONE register + immediates.
//   PAD 0 (dec.conv2, W = 56, OW = 55): pixel column 55 of a tile row is a dummy (its window column 56 is the next row's
//     column 0; results discarded, never stored);
//   PAD 1 (dec.conv4, W = 55, OW = 56): the window needs columns -1 .. 55 = 57; with pitch 56 the right padding column of
//     row r IS the left padding column of row r + 1 (both zero), and (5, 0) is the zero row behind the window.
//
// RT (round 4): the same kernel for ANY map size (BASELINE configs 4 / 5: 128 / 129 x 128 / 129 at 513 x 513, 199 / 200 x
// 303 / 304 at 800 x 1216).  A tile is four output rows x one SEGMENT of at most 55 output columns (OW_ = the segment
// pitch); W, OW and the segment count come from the arguments, a tile's window starts at the segment's first input column and
// everything beyond the image is out of range (zeros) -- with segments of <= 55 columns the 56-column window row holds every
// column both paddings need (no shared padding column).  Window pitch, tap immediates, fragment ring and epilogue are
// the static kernel's; what changes is the tile -> (image, row block, segment) split and two multiplications by OW.
template <int OW_, int PAD_, bool RT_ = false, int YP_ = 512>
struct Geo2 {
    static constexpr bool RT = RT_;
    static constexpr int YP = YP_;                               // bytes between output pixels: 512 = a 256-channel tensor of its own;
                                                                 // 1024: a 256-channel half of a 512-channel tensor (plain conv only)
    static constexpr int SEG = 55;                               // RT: output columns per segment
    static constexpr int OW = OW_, PAD = PAD_, W = OW_ + 1 - 2 * PAD_;
    static constexpr int ROWS = 4;                               // output rows per tile
    static constexpr int PWD = 56;                               // window row pitch = pixel-index pitch
    static constexpr int WROWS = (ROWS + 1) * PWD;               // 280 (+ the zero row (5, 0) for PAD 1: filled as out of range)
    static constexpr int NRG = (WROWS + 1 + 63) / 64;            // 64-row direct-to-LDS pieces per plane
    static constexpr int PLANE = NRG * 64 * 16;
    static constexpr int WIN_BYTES = 4 * PLANE;
    static constexpr int PX = ROWS * PWD;                        // 224 tile pixels (PAD 0: 4 of them dummies)
    static constexpr int MT = PX / 16;
    static constexpr int IMG0 = 2 * WIN_BYTES;                   // fused: bf16 image of the conv output, 32 planes
    static constexpr int IMG_PLANE = MT * 16 * 16;               // [224 rows][16 B]
    static constexpr int LDS_PLAIN = 2 * WIN_BYTES, LDS_FUSED = IMG0 + 32 * IMG_PLANE;
    static_assert(RT || (W + PAD <= PWD && OW <= PWD), "a window row holds the image row (+ one shared padding column)");
    static_assert(MT == 14 && NRG == 5 && PLANE % 256 == 0 && IMG_PLANE % 256 == 0, "14 row tiles, 5 pieces per plane");
    static_assert(WIN_BYTES + 13 * 256 + (PWD + 1) * 16 < 65536 && 16 * IMG_PLANE + 13 * 256 < 65536, "16-bit immediates");
    static_assert(LDS_FUSED <= 160 * 1024, "LDS");
};

constexpr int PF = 4;   // weight fragments are fetched this many k-steps ahead (= taps per slab: the ring slot of a k-step is its tap)
// vmcnt budget of a k-step's wait for its own two weight fragments (issue order, oldest first): ..., [its two loads], the six
// loads of the three k-steps behind it, and -- always inside those four k-steps -- the window pieces of one slab start (two
// or three per wave: two are counted).  Everything older than "the N youngest" has landed.
constexpr int VM_STEP = 2 * (PF - 1) + 2;

// One k-step: 14 pixel fragments x 2 weight fragments.  The fragment reads run seven row tiles ahead of the MFMAs through
// seven register quads: fragment i + 7 is read into the quad of fragment i as soon as its two MFMAs have been issued (28
// registers instead of 56: with all fourteen in flight the kernel spilled).  lgkmcnt retires in order: the wait before row
// tile i leaves min(13 - i, 6) younger reads outstanding.
#define SC2_W2_MMA_SEQ                                                                             \
    SC2_W2_MMA(0, 6) SC2_W2_MMA(1, 6) SC2_W2_MMA(2, 6) SC2_W2_MMA(3, 6) SC2_W2_MMA(4, 6) SC2_W2_MMA(5, 6) SC2_W2_MMA(6, 6) \
    SC2_W2_MMA(7, 6) SC2_W2_MMA(8, 5) SC2_W2_MMA(9, 4) SC2_W2_MMA(10, 3) SC2_W2_MMA(11, 2) SC2_W2_MMA(12, 1) SC2_W2_MMA(13, 0)

// pixel fragments of row tile i at a_base + OFF + 256 i (window planes, image planes alike); ABS: |.| on the fragment
// (norm GEMM of the fused GDN1); NVM: vmcnt budget of the wait for b0 / b1 (NVM_AFTER: ... when `after` is set: the first slab
// of a later tile, where the previous tile's output stores are younger than the fetch as well -- a branch around the wait
// only, so that both cases are ONE instruction stream: two copies of the slab made the copy audit path-blind)
//
// Round 4: the k-steps of one slab (and the eight of the norm GEMM) are CHAINED: while row tiles 7 .. 13 are multiplied, the
// quads they free take the NEXT step's fragments 0 .. 6 (OFF_NEXT relative to a_base_next), so that step starts with its reads
// in flight (PRE) and every wait is lgkmcnt(6).  Before, each step began by issuing seven reads and waiting for the first:
// with the two waves of a SIMD in lockstep behind the slab barrier nobody covered that latency (~150 - 250 cycles of every
// ~1 500-cycle step).  The chain stops at a slab boundary: the next slab's window is complete only behind its barrier.
// DEBUG builds only (-DSC2_W2_DBG_OUT=1: the fused kernels store x instead of y; 2: beta + norm).  As a RUNTIME argument the
// two tests sat in front of each of a tile's fourteen output stores: three scalar branches and sixteen register copies per row
// tile in the shipping kernel (round 4: found in the listing).
#ifndef SC2_W2_DBG_OUT
#define SC2_W2_DBG_OUT 0
#endif
#ifndef SC2_W2_ORDER
#define SC2_W2_ORDER 1   // 1: operand-stationary MFMA order inside a k-step (default since round 6); 0: b0 / b1 alternate on every MFMA (A/B)
#endif
#ifndef SC2_W2_CHAIN
#define SC2_W2_CHAIN 1   // 0: every k-step reads its own first seven fragments (A/B)
#endif
constexpr int NO_NEXT = -1;
template <int OFF, int NVM, bool ABS, int NVM_AFTER = NVM, bool PRE_ = false, int OFF_NEXT_ = NO_NEXT>
__device__ __forceinline__ void mma_step(f32x4_t (&acc)[14][2], uint32_t a_base, u32x4_t &b0, u32x4_t &b1, u32x4_t (&av)[7],
                                         bool after = false, uint32_t a_base_next = 0u) {
    constexpr bool PRE = SC2_W2_CHAIN && PRE_;
    constexpr int OFF_NEXT = SC2_W2_CHAIN ? OFF_NEXT_ : NO_NEXT;
    if constexpr (!PRE) {
#define SC2_W2_RD(i) av[i] = lds_read16_imm<OFF + (i) * 256>(a_base);
        SC2_W2_RD(0) SC2_W2_RD(1) SC2_W2_RD(2) SC2_W2_RD(3) SC2_W2_RD(4) SC2_W2_RD(5) SC2_W2_RD(6)
#undef SC2_W2_RD
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NVM_AFTER != NVM) {
        if (after) wait_vm<NVM_AFTER>();
        else wait_vm<NVM>();
    } else {
        wait_vm<NVM>();
    }
    const bf16x8_t bf0 = __builtin_bit_cast(bf16x8_t, b0), bf1 = __builtin_bit_cast(bf16x8_t, b1);
#if SC2_W2_ORDER
    // Operand-stationary order (round 6): seven MFMAs in a row share ONE weight fragment -- tiles 0 .. 6 with b0, the
    // same tiles with b1 (their quads are reloaded with tiles 7 .. 13 behind that second pass), then tiles 7 .. 13 likewise -- instead of
    // alternating b0 / b1 on every MFMA.  Same products into the same accumulators in the same k order: bit-identical results.  These
    // launches are POWER-limited on real data (tools/clock_probe.py --zero-data: 1.85 -> 2.33 GHz at the same pipe share on zeros), and an
    // operand that does not change between MFMAs toggles less: dec.conv2 + IGDN256 0.715 -> 0.706 ms, dec.conv2 0.644 -> 0.639, alternating
    // A/B on one box (profiles/r06y_mfma_order_ab.txt); dec.conv4 and the tail form unchanged.
#define SC2_W2_PASS_A(i, NWAIT)                                                                 \
    {                                                                                           \
        wait_lgkm<NWAIT>(av[(i) % 7]);                                                          \
        const u32x4_t m = ABS ? av[(i) % 7] & 0x7FFF7FFFu : av[(i) % 7];                        \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, __builtin_bit_cast(bf16x8_t, m), acc[i][0], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
#define SC2_W2_PASS_B(i)                                                                        \
    {                                                                                           \
        const u32x4_t m = ABS ? av[(i) % 7] & 0x7FFF7FFFu : av[(i) % 7];                        \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, __builtin_bit_cast(bf16x8_t, m), acc[i][1], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if constexpr ((i) + 7 < 14) av[(i) % 7] = lds_read16_imm<OFF + ((i) + 7 < 14 ? (i) + 7 : 0) * 256>(a_base); \
        else if constexpr (OFF_NEXT != NO_NEXT)                                                 \
            av[(i) % 7] = lds_read16_imm<(OFF_NEXT != NO_NEXT ? OFF_NEXT : 0) + ((i) >= 7 ? (i) - 7 : 0) * 256>(a_base_next); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
    // (waits: nothing is read during a pass A, so tile i's wait leaves the 6 - i / 13 - i younger reads of its half outstanding)
    SC2_W2_PASS_A(0, 6) SC2_W2_PASS_A(1, 5) SC2_W2_PASS_A(2, 4) SC2_W2_PASS_A(3, 3) SC2_W2_PASS_A(4, 2) SC2_W2_PASS_A(5, 1) SC2_W2_PASS_A(6, 0)
    SC2_W2_PASS_B(0) SC2_W2_PASS_B(1) SC2_W2_PASS_B(2) SC2_W2_PASS_B(3) SC2_W2_PASS_B(4) SC2_W2_PASS_B(5) SC2_W2_PASS_B(6)
    SC2_W2_PASS_A(7, 6) SC2_W2_PASS_A(8, 5) SC2_W2_PASS_A(9, 4) SC2_W2_PASS_A(10, 3) SC2_W2_PASS_A(11, 2) SC2_W2_PASS_A(12, 1) SC2_W2_PASS_A(13, 0)
    SC2_W2_PASS_B(7) SC2_W2_PASS_B(8) SC2_W2_PASS_B(9) SC2_W2_PASS_B(10) SC2_W2_PASS_B(11) SC2_W2_PASS_B(12) SC2_W2_PASS_B(13)
#undef SC2_W2_PASS_A
#undef SC2_W2_PASS_B
#else
#define SC2_W2_MMA(i, NWAIT)                                                                    \
    {                                                                                           \
        wait_lgkm<(OFF_NEXT != NO_NEXT) ? 6 : NWAIT>(av[(i) % 7]);                              \
        const u32x4_t m = ABS ? av[(i) % 7] & 0x7FFF7FFFu : av[(i) % 7];                        \
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, m);                                    \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, af, acc[i][0], 0, 0, 0);       \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, af, acc[i][1], 0, 0, 0);       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if constexpr ((i) + 7 < 14) av[(i) % 7] = lds_read16_imm<OFF + ((i) + 7 < 14 ? (i) + 7 : 0) * 256>(a_base); \
        else if constexpr (OFF_NEXT != NO_NEXT)                                                 \
            av[(i) % 7] = lds_read16_imm<(OFF_NEXT != NO_NEXT ? OFF_NEXT : 0) + ((i) >= 7 ? (i) - 7 : 0) * 256>(a_base_next); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
    SC2_W2_MMA_SEQ
#undef SC2_W2_MMA
#endif
}
#undef SC2_W2_MMA_SEQ

// MODE 0: conv;  1: conv + (inverse) GDN1;  2: conv + the two 1x1 layers of the caller that read its output ("tail")
// k-step of the tail's stride-2 1x1 layer: 4 row tiles = the tile's 56 pixels with even row and column (per-lane image rows)
template <int OFF, int NVM>
__device__ __forceinline__ void ds_step(f32x4_t (&acc)[14][2], const uint32_t (&base)[4], u32x4_t &b0, u32x4_t &b1) {
    u32x4_t av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = lds_read16_imm<OFF>(base[i]);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<NVM>();
    const bf16x8_t bf0 = __builtin_bit_cast(bf16x8_t, b0), bf1 = __builtin_bit_cast(bf16x8_t, b1);
#define SC2_W2_DS(i)                                                                            \
    {                                                                                           \
        wait_lgkm<3 - (i)>(av[i]);                                                              \
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[i]);                                \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0, af, acc[i][0], 0, 0, 0);       \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1, af, acc[i][1], 0, 0, 0);       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
    SC2_W2_DS(0) SC2_W2_DS(1) SC2_W2_DS(2) SC2_W2_DS(3)
#undef SC2_W2_DS
}

template <class G, int MODE, bool INVERSE>
__global__ __launch_bounds__(512, 2) void conv2x2_win_kernel(const W2Args p) {
    static_assert(MODE == 0 || G::YP == 512, "the fused / tail forms write 256-channel tensors of their own");
    constexpr int MT = G::MT, PAD = G::PAD, PWD = G::PWD;
    constexpr bool FUSE = MODE == 1, TAIL = MODE == 2, RT = G::RT;
    static_assert(!(RT && TAIL), "the tail mode is the 224 x 224 geometry's");
    const int W = RT ? p.W : G::W, OW = RT ? p.OW : G::OW;       // (compile-time constants unless RT)
    // tile -> image, first output row, first output column and width of its segment
    auto tile_origin = [&](int tile, int &img, int &oh0, int &ow0, int &sw) {
        img = tile / p.tiles_per_img;
        const int t_in = tile - img * p.tiles_per_img;
        if constexpr (RT) {
            const int rb = t_in / p.nseg, seg = t_in - rb * p.nseg;
            oh0 = rb * G::ROWS;
            ow0 = seg * G::SEG;
            sw = OW - ow0 < G::SEG ? OW - ow0 : G::SEG;
        } else {
            oh0 = t_in * G::ROWS;
            ow0 = 0;
            sw = OW;
        }
    };
    constexpr uint32_t OOB = 0x80000000u;
    // output stores per lane and tile, ALWAYS issued (masked ones out of range): the counted vmcnt waits of the next tile rely
    // on it.  Tail: o1, y (out of range as a whole when the caller does not want y), 2 x 4 ods.
    constexpr int NSTORE = TAIL ? 2 * MT + 8 : MT;
    constexpr int NSTORE_LAST = TAIL ? 4 : MT;   // ... of them behind the last weight fetch of the tile
    // The next tile's first four k-steps are fetched during the last four k-steps of this one, i.e. across the epilogue -- except
    // in the forward-GDN1 variant, whose divisions need the registers: there the allocator SPILLED the fragments in flight
    // (tools/audit_vmcnt.py --copies), so that variant fetches them behind its output stores (and waits for those stores'
    // acknowledgements with them: it is not on the bottleneck's path, which uses the inverse form).
    constexpr bool WRAP = !(FUSE && !INVERSE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cin = p.Cin, H = p.H, OH = p.OH;
    const int NS = Cin >> 5;                 // 32-channel slabs (even)
    const uint32_t KT = (uint32_t)NS * 4u;   // conv k-steps
    const uint32_t KTT = KT + (FUSE ? 8u : TAIL ? 24u : 0u);

    // this workgroup's tiles: XCD x (blockIdx & 7) owns a contiguous range of the output, cut into runs of tiles_per_wg
    const int slot = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int t_first = slot * p.tiles_per_wg;
    const int t_last = t_first + p.tiles_per_wg < p.n_tiles ? t_first + p.tiles_per_wg : p.n_tiles;
    if (t_first >= t_last) return;

    const buf_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
    const i32x4_t rs_w = rsrc_words(p.w, p.w_bytes);
    const buf_rsrc_t rs_y = make_rsrc(p.y, p.y_bytes);
    [[maybe_unused]] const buf_rsrc_t rs_o1 = make_rsrc(TAIL ? p.o1 : p.y, TAIL ? p.o1_bytes : 0u);
    [[maybe_unused]] const buf_rsrc_t rs_ods = make_rsrc(TAIL ? p.ods : p.y, TAIL ? p.ods_bytes : 0u);

    // window fill: 20 pieces per slab (4 chunk planes x 5 row groups of 64): wave w fills plane w & 3, row groups {0, 1, 2}
    // (waves 0-3) or {3, 4} (waves 4-7)
    const int pq = wave & 3, pj0 = wave < 4 ? 0 : 3, pn = wave < 4 ? 3 : 2;
    uint32_t pw_vo[3];
    auto window_offsets = [&](int tile, bool live) {   // live = false: every lane out of range (zeros; see SC2_W2_SLAB)
        int img, oh0, ow0, sw;
        tile_origin(tile, img, oh0, ow0, sw);
        // (the lane index is re-derived here and the values below recomputed per tile: hoisted out of the tile loop they were spilled)
        int ln;   // (volatile: computed where it is used)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int wr = (pj0 + j) * 64 + ln;
            // wr / 56 and wr % 56 for wr < 320, by multiply-shift and shifts (a 32-bit `wr - ihp * 56` became v_mad_u64_u32 with
            // a don't-care high addend register -- which the copy audit cannot tell from a read of a fragment in flight)
            static_assert(PWD == 56 && G::NRG * 64 <= 320, "ihp = (wr * 1171) >> 16 is exact below 336");
            const int ihp = (int)(((uint32_t)wr * 1171u) >> 16), iwp = wr - ((ihp << 6) - (ihp << 3));
            const int ih = oh0 - PAD + ihp, iw = ow0 + iwp - PAD;
            const bool ok = live & (j < pn) & (wr < G::WROWS) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            pw_vo[j] = ok ? (uint32_t)((((img * H + ih) * W + iw) * Cin) * 2 + pq * 16) : OOB;
        }
    };
    auto issue_window = [&](int cb, int par) {
        unsigned char *dst = smem + par * G::WIN_BYTES + pq * G::PLANE + pj0 * 1024;
        buf_load_lds16(rs_x, (lds_ptr_t)(dst), pw_vo[0], (uint32_t)cb * 64u);
        buf_load_lds16(rs_x, (lds_ptr_t)(dst + 1024), pw_vo[1], (uint32_t)cb * 64u);
        if (pn == 3) buf_load_lds16(rs_x, (lds_ptr_t)(dst + 2048), pw_vo[2], (uint32_t)cb * 64u);
    };

    // fragment row of this lane in row tile 0 at tap (0, 0); row tile i: + 256 i, tap (kh, kw): + (kh PWD + kw) 16
    const uint32_t a_base = lds_base + (uint32_t)(fq * G::PLANE + frow * 16);
    // weights: k-step k, 16-channel tile t -> 1 KB at (k * 16 + t) * 1024; this wave's tiles are 2 w, 2 w + 1
    const uint32_t b_vo = (uint32_t)(lane * 16);
    const uint32_t b_so0 = (uint32_t)(2 * wave) * 1024u;
    u32x4_t bq[PF][2];
    auto fetch_b = [&](uint32_t k, u32x4_t &b0, u32x4_t &b1) {   // k in [0, 2 KTT): wraps to the next tile's first k-steps
        const uint32_t kk = k >= KTT ? k - KTT : k;
        const uint32_t so = b_so0 + kk * 16384u;
        wload16(b0, rs_w, b_vo, so);
        wload16(b1, rs_w, b_vo, so + 1024u);
    };

    f32x4_t acc[MT][2];
    u32x4_t av[7];   // the fragment ring (mma_step): lives across the chained k-steps

    window_offsets(t_first, true);
    issue_window(0, 0);
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch_b((uint32_t)s, bq[s][0], bq[s][1]);

    for (int tile = t_first; tile < t_last; ++tile) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const bool first = tile == t_first;

        // k-step TAP of slab cb: its fragments were fetched four k-steps ago; the fetch of k-step + 4 goes into the same
        // registers behind the step's last MFMA.  In the first slab of a later tile (SLAB0 && !first) the previous tile's last
        // NSTORE_LAST output stores are younger than the fetch as well.
#define SC2_W2_STEP(PAR, cb, TAP, SLAB0)                                                                                \
    {                                                                                                                   \
        mma_step<PAR * G::WIN_BYTES + ((TAP / 2) * PWD + TAP % 2) * 16, VM_STEP, false,                                  \
                 VM_STEP + (SLAB0 && WRAP && !TAIL ? NSTORE_LAST : 0), (TAP > 0),                                        \
                 (TAP < 3 ? PAR * G::WIN_BYTES + (((TAP + 1) / 2) * PWD + (TAP + 1) % 2) * 16 : NO_NEXT)>(               \
            acc, a_base, bq[TAP][0], bq[TAP][1], av, !first, a_base);                                                    \
        fetch_b((uint32_t)(cb) * 4u + (TAP + PF), bq[TAP][0], bq[TAP][1]);                                               \
    }
#define SC2_W2_SLAB(PAR, cb, SLAB0)                                                                                     \
    {                                                                                                                   \
        /* this wave's share of the slab's window has landed: it is older than the 2 PF weight loads in flight and, in \
           the first slab of a later tile, than the previous tile's output stores */                                \
        if (SLAB0 && !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PF + NSTORE) : "memory");                     \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PF) : "memory");                                              \
        __builtin_amdgcn_s_barrier();   /* window complete; everybody is done with the previous slab's window */    \
        /* EVERY slab start issues this wave's two or three window pieces (VM_STEP counts two of them): the next slab's, \
           or the next tile's first window (buffer 0: NS is even), or -- behind the workgroup's last tile -- zeros into the \
           dead buffer 0 */                                                                                         \
        if ((cb) + 1 < NS) {                                                                                        \
            issue_window((cb) + 1, 1 - PAR);                                                                        \
        } else {                                                                                                    \
            window_offsets(tile + 1 < t_last ? tile + 1 : tile, tile + 1 < t_last);                                 \
            issue_window(0, 0);                                                                                     \
        }                                                                                                           \
        SC2_W2_STEP(PAR, cb, 0, SLAB0) SC2_W2_STEP(PAR, cb, 1, SLAB0) SC2_W2_STEP(PAR, cb, 2, SLAB0)                    \
        SC2_W2_STEP(PAR, cb, 3, SLAB0)                                                                                  \
    }
        // the first two slabs peeled (the first one's waits branch on `first`), then the steady state
        SC2_W2_SLAB(0, 0, true)
        SC2_W2_SLAB(1, 1, false)
        for (int cb = 2; cb < NS; cb += 2) {
            SC2_W2_SLAB(0, cb, false)
            SC2_W2_SLAB(1, cb + 1, false)
        }
#undef SC2_W2_SLAB
#undef SC2_W2_STEP

        // ---- output.  Lane (frow, fq) holds, for row tile i, channels 32 w + 8 fq + [0, 4) in acc[i][0] and + [4, 8) in
        // acc[i][1] (the packing permutes the weight rows that way) of tile pixel 16 i + frow = (row ml / 56, column ml % 56)
        // Stores go through a buffer descriptor with invalid lanes sent out of range: ALWAYS 14 store instructions per tile,
        // which the counted vmcnt wait of the next tile's first slab relies on.
        int img, oh0, ow0, sw;
        tile_origin(tile, img, oh0, ow0, sw);
        const int rows_valid = OH - oh0 < G::ROWS ? OH - oh0 : G::ROWS;
        const uint32_t y_so = (uint32_t)((img * OH + oh0) * OW + ow0) * (uint32_t)G::YP;   // tile base (bytes), scalar
        int ln_o;   // the lane index, computed HERE (volatile): the per-row-tile offsets and masks derived from it are then recomputed per
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_o));   // tile, not hoisted and spilled
        const int fr = ln_o & 15, fqo = ln_o >> 4;
        const uint32_t y_ch = (uint32_t)((32 * wave + 8 * fqo) * 2);
        // byte offset of tile pixel ml = 16 i + fr in the tile's block of y (pixel pitch `pitch` bytes), or out of range
        // (store groups that run behind further k-steps re-derive the lane index where they run -- lane_now(): offsets
        //  computed from ONE copy were all formed up front and held, or spilled, across those steps)
        auto lane_now = []() {
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            return l;
        };
        auto out_off = [&](int i, int fr_, uint32_t ch, int pitch) -> uint32_t {
            const int ml = i * 16 + fr_;
            const int r = (ml >= PWD) + (ml >= 2 * PWD) + (ml >= 3 * PWD), c = ml - r * PWD;
            const bool ok = (r < rows_valid) & (c < sw);
            return ok ? (uint32_t)((r * OW + c) * pitch) + ch : OOB;
        };
        if constexpr (FUSE || TAIL) {
            // image slots (computed here, per tile, from the re-derived lane index: held across the K loop they were spilled).
            // Writer / read-back: plane 4 wave + fq, row 16 i + frow; second-GEMM reader: plane 4 s + fq
            const uint32_t img_wr = lds_base + (uint32_t)(G::IMG0 + (4 * wave + fqo) * G::IMG_PLANE + fr * 16);
            const uint32_t g_base0 = lds_base + (uint32_t)(G::IMG0 + fqo * G::IMG_PLANE + fr * 16);
            const uint32_t g_base1 = g_base0 + 16u * G::IMG_PLANE;
            // x -> bf16 image
#define SC2_W2_WR(i)                                                                                                        \
    lds_write16_imm<(i) * 256>(img_wr, u32x4_t{pack2(acc[i][0][0], acc[i][0][1]), pack2(acc[i][0][2], acc[i][0][3]),      \
                                               pack2(acc[i][1][0], acc[i][1][1]), pack2(acc[i][1][2], acc[i][1][3])});
            SC2_W2_WR(0) SC2_W2_WR(1) SC2_W2_WR(2) SC2_W2_WR(3) SC2_W2_WR(4) SC2_W2_WR(5) SC2_W2_WR(6)
            SC2_W2_WR(7) SC2_W2_WR(8) SC2_W2_WR(9) SC2_W2_WR(10) SC2_W2_WR(11) SC2_W2_WR(12) SC2_W2_WR(13)
#undef SC2_W2_WR
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // the image is complete
            // epilogue constants of this lane's eight channels (beta, or the tail's bias1): asm loads as well -- a compiler-tracked
            // load here made its use wait vmcnt(0), i.e. for the next tile's weight fragments fetched just before.  Issued in
            // front of the eight norm k-steps (whose budgets do not count them: two more young loads only make those waits
            // reach one fetch further back), waited for behind them.
            const i32x4_t rs_c0 = rsrc_words(TAIL ? p.bias1 : p.beta, TAIL ? 128u * 4u : 256u * 4u);
            u32x4_t c0_lo, c0_hi;
            {
                const uint32_t co = (uint32_t)(((TAIL ? (wave < 4 ? 32 * wave : 0) : 32 * wave) + 8 * fqo) * 4);
                cload16(c0_lo, rs_c0, co, 0u);
                cload16(c0_hi, rs_c0, co, 16u);
            }
            // norm = gamma |x|: k-steps KT .. KT + 7 of the weight stream, ring slot = step & 3 (no window piece and no store
            // is issued between such a step and its fetch: six younger loads)
#define SC2_W2_NSTEP(S, BASE, OFF)                                                         \
    {                                                                                      \
        constexpr int nvm = (!WRAP && (S) >= 4) ? 2 * (7 - (S)) : 2 * (PF - 1);            \
        mma_step<OFF, nvm, !TAIL, nvm, ((S) > 0), ((S) < 7 ? (((S) + 1) & 3) * 4 * G::IMG_PLANE : NO_NEXT)>(                  \
            acc, BASE, bq[(S) & 3][0], bq[(S) & 3][1], av, false, (S) + 1 < 4 ? g_base0 : g_base1);                            \
        if constexpr (WRAP || (S) < 4) fetch_b(KT + (S) + PF, bq[(S) & 3][0], bq[(S) & 3][1]);                                \
    }
            SC2_W2_NSTEP(0, g_base0, 0 * G::IMG_PLANE) SC2_W2_NSTEP(1, g_base0, 4 * G::IMG_PLANE)
            SC2_W2_NSTEP(2, g_base0, 8 * G::IMG_PLANE) SC2_W2_NSTEP(3, g_base0, 12 * G::IMG_PLANE)
            SC2_W2_NSTEP(4, g_base1, 0 * G::IMG_PLANE) SC2_W2_NSTEP(5, g_base1, 4 * G::IMG_PLANE)
            SC2_W2_NSTEP(6, g_base1, 8 * G::IMG_PLANE) SC2_W2_NSTEP(7, g_base1, 12 * G::IMG_PLANE)
#undef SC2_W2_NSTEP
            if constexpr (TAIL) {
                // ---- o1 = relu(W1 y + bias1): waves 0-3 hold channels 32 w + 8 fq + [0, 8) (the stream's rows 128-255 are zero);
                // waves 4-7 issue the same 14 stores out of range (a fixed number of stores per wave and tile)
                {
                    const bool mine = wave < 4;
                    wait_vm_tied<2 * PF>(c0_lo, c0_hi);   // (sixteen younger fetches)
                    const float4 b_lo = __builtin_bit_cast(float4, c0_lo), b_hi = __builtin_bit_cast(float4, c0_hi);
                    const uint32_t o_so = (uint32_t)((img * OH + oh0) * OW) * 256u;
                    const int l1 = lane_now();
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const float r[8] = {fmaxf(acc[i][0][0] + b_lo.x, 0.f), fmaxf(acc[i][0][1] + b_lo.y, 0.f), fmaxf(acc[i][0][2] + b_lo.z, 0.f),
                                            fmaxf(acc[i][0][3] + b_lo.w, 0.f), fmaxf(acc[i][1][0] + b_hi.x, 0.f), fmaxf(acc[i][1][1] + b_hi.y, 0.f),
                                            fmaxf(acc[i][1][2] + b_hi.z, 0.f), fmaxf(acc[i][1][3] + b_hi.w, 0.f)};
                        const uint32_t vo = out_off(i, l1 & 15, (uint32_t)((32 * wave + 8 * (l1 >> 4)) * 2), 256);
                        buf_store16(rs_o1, mine ? vo : OOB, o_so, u32x4_t{pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7])});
                    }
                }
                // ---- y itself (read back from the image: the accumulators are gone).  A caller that does not want it passes a
                // null y with a zero-sized descriptor: the 14 stores are issued all the same and dropped (the vmcnt budgets
                // count them)
                {
                    u32x4_t xr[MT];
#define SC2_W2_RD(i) xr[i] = lds_read16_imm<(i) * 256>(img_wr);
                    SC2_W2_RD(0) SC2_W2_RD(1) SC2_W2_RD(2) SC2_W2_RD(3) SC2_W2_RD(4) SC2_W2_RD(5) SC2_W2_RD(6)
                    SC2_W2_RD(7) SC2_W2_RD(8) SC2_W2_RD(9) SC2_W2_RD(10) SC2_W2_RD(11) SC2_W2_RD(12) SC2_W2_RD(13)
#undef SC2_W2_RD
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]),
                                 "+v"(xr[7]), "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13])::"memory");
                    const int l2 = lane_now();
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        buf_store16(rs_y, out_off(i, l2 & 15, (uint32_t)((32 * wave + 8 * (l2 >> 4)) * 2), 512), y_so, xr[i]);
                }
                // ---- ods = Wds y[::2, ::2] + bias_ds: two passes of 256 output channels over the tile's 56 even pixels
                const uint32_t d_so = (uint32_t)((img * (OH / 2) + oh0 / 2) * (OW / 2)) * 1024u;
                uint32_t ds_lo[4], ds_hi[4];   // image rows of the even pixels (row 2 r', column 2 c'); computed here, per tile: held
                const int l3 = lane_now();             // across the K loop they were spilled
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int m = i * 16 + (l3 & 15);
                    m = m < 2 * (OW / 2) ? m : 0;
                    const int r2 = m / (OW / 2), c2 = m - r2 * (OW / 2);
                    ds_lo[i] = lds_base + (uint32_t)(G::IMG0 + (l3 >> 4) * G::IMG_PLANE + (2 * r2 * PWD + 2 * c2) * 16);
                    ds_hi[i] = ds_lo[i] + 16u * G::IMG_PLANE;
                }
                // vmcnt budgets: the first four k-steps of a pass were fetched in front of the stores issued since (pass 0: o1 + y,
                // 2 MT; pass 1: the four ods stores of pass 0)
#define SC2_W2_DSTEP(S, BASE, OFF, NVM)                                                    \
    {                                                                                      \
        ds_step<OFF, NVM>(acc, BASE, bq[(S) & 3][0], bq[(S) & 3][1]);                      \
        fetch_b(KT + (S) + PF, bq[(S) & 3][0], bq[(S) & 3][1]);                            \
    }
#define SC2_W2_DPASS(PASS, NVM0)                                                                                             \
    {                                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};                                                                         \
            acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};                                                                         \
        }                                                                                                                    \
        u32x4_t cd_lo, cd_hi;   /* bias_ds of this pass: asm loads in front of its eight k-steps, waited for behind them */          \
        const int l4 = lane_now();                                                                                          \
        cload16(cd_lo, rs_cd, (uint32_t)((256 * PASS + 32 * wave + 8 * (l4 >> 4)) * 4), 0u);                                      \
        cload16(cd_hi, rs_cd, (uint32_t)((256 * PASS + 32 * wave + 8 * (l4 >> 4)) * 4), 16u);                                     \
        SC2_W2_DSTEP(8 + 8 * PASS + 0, ds_lo, 0 * G::IMG_PLANE, NVM0) SC2_W2_DSTEP(8 + 8 * PASS + 1, ds_lo, 4 * G::IMG_PLANE, NVM0)   \
        SC2_W2_DSTEP(8 + 8 * PASS + 2, ds_lo, 8 * G::IMG_PLANE, NVM0) SC2_W2_DSTEP(8 + 8 * PASS + 3, ds_lo, 12 * G::IMG_PLANE, NVM0)  \
        SC2_W2_DSTEP(8 + 8 * PASS + 4, ds_hi, 0 * G::IMG_PLANE, 2 * (PF - 1)) SC2_W2_DSTEP(8 + 8 * PASS + 5, ds_hi, 4 * G::IMG_PLANE, 2 * (PF - 1))   \
        SC2_W2_DSTEP(8 + 8 * PASS + 6, ds_hi, 8 * G::IMG_PLANE, 2 * (PF - 1)) SC2_W2_DSTEP(8 + 8 * PASS + 7, ds_hi, 12 * G::IMG_PLANE, 2 * (PF - 1)) \
        wait_vm_tied<2 * PF>(cd_lo, cd_hi);                                                                                       \
        const float4 b_lo = __builtin_bit_cast(float4, cd_lo), b_hi = __builtin_bit_cast(float4, cd_hi);                     \
        const int l5 = lane_now();                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            const int m2 = i * 16 + (l5 & 15);                                                                               \
            const int r2 = m2 / (OW / 2), c2 = m2 - r2 * (OW / 2);                                                           \
            const bool ok = (m2 < 2 * (OW / 2)) & (2 * r2 < rows_valid);                                                     \
            const uint32_t vo = (uint32_t)((r2 * (OW / 2) + c2) * 1024 + (256 * PASS + 32 * wave + 8 * (l5 >> 4)) * 2);      \
            buf_store16(rs_ods, ok ? vo : OOB, d_so,                                                                         \
                        u32x4_t{pack2(acc[i][0][0] + b_lo.x, acc[i][0][1] + b_lo.y), pack2(acc[i][0][2] + b_lo.z, acc[i][0][3] + b_lo.w), \
                                pack2(acc[i][1][0] + b_hi.x, acc[i][1][1] + b_hi.y), pack2(acc[i][1][2] + b_hi.z, acc[i][1][3] + b_hi.w)}); \
        }                                                                                                                    \
    }
                const i32x4_t rs_cd = rsrc_words(p.bias_ds, 512u * 4u);
                SC2_W2_DPASS(0, 2 * (PF - 1) + 2 * MT)
                SC2_W2_DPASS(1, 2 * (PF - 1) + 4)
#undef SC2_W2_DPASS
#undef SC2_W2_DSTEP
            } else {
                // y = x * (beta + norm)  (inverse)  or  x / (beta + norm); x from this lane's own image slots.  (beta is fetched per
                // tile: 32 bytes per lane out of L2; kept in registers across the K loop it was spilled to scratch)
                wait_vm_tied<2 * PF>(c0_lo, c0_hi);   // (sixteen younger fetches)
                const float4 beta_lo = __builtin_bit_cast(float4, c0_lo), beta_hi = __builtin_bit_cast(float4, c0_hi);
                // (two batches of seven row tiles: with all fourteen read-backs live the forward-GDN form, whose divisions need more
                //  temporaries, spilled)
#define SC2_W2_FIN(i0)                                                                                                              \
    {                                                                                                                               \
        u32x4_t xr[7];                                                                                                              \
        xr[0] = lds_read16_imm<((i0) + 0) * 256>(img_wr); xr[1] = lds_read16_imm<((i0) + 1) * 256>(img_wr);                         \
        xr[2] = lds_read16_imm<((i0) + 2) * 256>(img_wr); xr[3] = lds_read16_imm<((i0) + 3) * 256>(img_wr);                         \
        xr[4] = lds_read16_imm<((i0) + 4) * 256>(img_wr); xr[5] = lds_read16_imm<((i0) + 5) * 256>(img_wr);                         \
        xr[6] = lds_read16_imm<((i0) + 6) * 256>(img_wr);                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                         \
                     : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6])::"memory");         \
        _Pragma("unroll") for (int k = 0; k < 7; ++k) {                                                                             \
            const int i = (i0) + k;                                                                                                 \
            const float xv[8] = {__builtin_bit_cast(float, xr[k][0] << 16), __builtin_bit_cast(float, xr[k][0] & 0xFFFF0000u),      \
                                 __builtin_bit_cast(float, xr[k][1] << 16), __builtin_bit_cast(float, xr[k][1] & 0xFFFF0000u),      \
                                 __builtin_bit_cast(float, xr[k][2] << 16), __builtin_bit_cast(float, xr[k][2] & 0xFFFF0000u),      \
                                 __builtin_bit_cast(float, xr[k][3] << 16), __builtin_bit_cast(float, xr[k][3] & 0xFFFF0000u)};     \
            const float nm[8] = {beta_lo.x + acc[i][0][0], beta_lo.y + acc[i][0][1], beta_lo.z + acc[i][0][2], beta_lo.w + acc[i][0][3], \
                                 beta_hi.x + acc[i][1][0], beta_hi.y + acc[i][1][1], beta_hi.z + acc[i][1][2], beta_hi.w + acc[i][1][3]}; \
            float r[8];                                                                                                             \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) r[e] = INVERSE ? xv[e] * nm[e] : xv[e] * (1.0f / nm[e]);                  \
            if (SC2_W2_DBG_OUT == 1) {                                                                                              \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) r[e] = xv[e];                                                         \
            } else if (SC2_W2_DBG_OUT == 2) {                                                                                       \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) r[e] = nm[e];                                                         \
            }                                                                                                                       \
            buf_store16(rs_y, out_off(i, fr, y_ch, 512), y_so,                                                                      \
                        u32x4_t{pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7])});                       \
        }                                                                                                                           \
    }
                SC2_W2_FIN(0)
                SC2_W2_FIN(7)
#undef SC2_W2_FIN
                if constexpr (!WRAP) {   // (see WRAP: the next tile's first four k-steps, as in the prologue)
#pragma unroll
                    for (int s4 = 0; s4 < PF; ++s4) fetch_b((uint32_t)s4, bq[s4][0], bq[s4][1]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                buf_store16(rs_y, out_off(i, fr, y_ch, G::YP), y_so,
                            u32x4_t{pack2(acc[i][0][0], acc[i][0][1]), pack2(acc[i][0][2], acc[i][0][3]),
                                    pack2(acc[i][1][0], acc[i][1][1]), pack2(acc[i][1][2], acc[i][1][3])});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight fragments prefetched for a tile that does not exist
}

template <class G, int MODE, bool INVERSE = true>
int launch_w2(W2Args a, hipStream_t s) {
    constexpr int LDS = MODE != 0 ? G::LDS_FUSED : G::LDS_PLAIN;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2x2_win_kernel<G, MODE, INVERSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS);
        attr_set = true;
    }
    const int g_cus_w2 = sc2_device_cus();
    a.nseg = G::RT ? (a.OW + G::SEG - 1) / G::SEG : 1;
    a.tiles_per_img = (a.OH + G::ROWS - 1) / G::ROWS * a.nseg;
    a.n_tiles = a.N * a.tiles_per_img;
    // runs of `tiles_per_wg` consecutive tiles per workgroup.  Default 2: many short workgroups that the dispatcher places on
    // whatever CUs are free -- inside the pipelined bench the serial coder's workgroups and the encoder stage of a later batch
    // hold CUs for milliseconds, and with one static share per CU (14 tiles at bs 256) the launch waited for the workgroups
    // that could not start (measured in the pipeline, K = 20: 0.85 ms with the whole share, 0.79 ms with runs of 2 or 1;
    // stand-alone the same).  SC2_W2_RUN=<tiles> overrides (tools/attic/w2_run_ab.sh).
    const int cus8 = (g_cus_w2 / 8) * 8 > 0 ? (g_cus_w2 / 8) * 8 : 8;
    const int share = (a.n_tiles + cus8 - 1) / cus8;
    int run = share < 2 ? share : 2;
    if (sc2_pol().w2_run > 0) run = sc2_pol().w2_run;
    a.tiles_per_wg = run;
    int grid = (a.n_tiles + run - 1) / run;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((conv2x2_win_kernel<G, MODE, INVERSE>), dim3(grid), dim3(512), LDS, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

typedef Geo2<55, 0> Gd2;   // dec.conv2: 56 -> 55
typedef Geo2<56, 1> Gd4;   // dec.conv4: 55 -> 56
typedef Geo2<55, 0, true> Gr0;   // any width, pad 0 (W -> W - 1)
typedef Geo2<55, 1, true> Gr1;   // any width, pad 1 (W -> W + 1)
typedef Geo2<56, 1, false, 1024> Gd4w;   // 55 -> 56 into one 256-channel half of a 512-channel tensor (data gradient of dec.conv2)
typedef Geo2<55, 1, true, 1024> Gr1w;

}  // namespace

extern "C" int sc2_conv2x2_win_supported(int H, int W, int Cin, int Cout, int pad) {
    if (Cout != 256 || Cin < 64 || Cin % 64 != 0) return 0;
    if (pad == 0) return W >= 2 && H >= 2 ? 1 : 0;   // (56: the static geometry; any other width: segments of 55 columns)
    if (pad == 1) return W >= 1 && H >= 1 ? 1 : 0;   // (55: static)
    return 0;
}

extern "C" int sc2_conv2x2_win_fwd(const void *x, const void *w_frag, const float *beta, void *y, int N, int H, int W, int Cin,
                                   int pad, int fused, int inverse, int y_channels, int y_channel0, void *stream) {
    SC2_REQUIRE(x && w_frag && y, SC2_ERR_INVALID_ARG, "conv2x2_win: null argument");
    SC2_REQUIRE(N > 0, SC2_ERR_INVALID_ARG, "conv2x2_win: non-positive batch");
    SC2_REQUIRE(sc2_conv2x2_win_supported(H, W, Cin, 256, pad), SC2_ERR_UNSUPPORTED,
                "conv2x2_win: needs Cout 256, Cin %% 64 == 0, pad 0 or 1 and a map of at least 2 x 2 (pad 0); got %d x %d, Cin %d, pad %d",
                H, W, Cin, pad);
    SC2_REQUIRE(!fused || beta, SC2_ERR_INVALID_ARG, "conv2x2_win: the fused GDN1 needs beta");
    const int ksteps = Cin / 32 * 4 + (fused ? 8 : 0);
    const long long x_bytes = (long long)N * H * W * Cin * 2, w_bytes = (long long)ksteps * 16384;
    // y_channels 512: this launch's 256 output channels are channels [y_channel0, y_channel0 + 256) of a 512-channel tensor (the
    // data gradient of a 512 -> 256 layer is two such launches); plain pad-1 convolutions only
    const bool wide = y_channels == 512;
    SC2_REQUIRE(y_channels == 256 ? y_channel0 == 0 : (wide && (y_channel0 == 0 || y_channel0 == 256) && !fused && pad == 1),
                SC2_ERR_UNSUPPORTED, "conv2x2_win: output channels %d at offset %d (256 at 0, or a half of 512 for a plain pad-1 conv)",
                y_channels, y_channel0);
    const long long y_bytes = (long long)N * (H + 2 * pad - 1) * (W + 2 * pad - 1) * (wide ? 1024 : 512) - y_channel0 * 2;
    SC2_REQUIRE(x_bytes < 0x7FF00000LL && y_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv2x2_win: tensor of %lld bytes exceeds 2 GB",
                x_bytes > y_bytes ? x_bytes : y_bytes);
    W2Args a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.beta = beta;
    a.y = static_cast<uint16_t *>(y) + y_channel0;
    a.N = N; a.H = H; a.Cin = Cin; a.OH = H + 2 * pad - 1;
    a.W = W; a.OW = W + 2 * pad - 1; a.nseg = 1;
    a.n_tiles = 0; a.tiles_per_img = 0; a.tiles_per_wg = 0;
    a.x_bytes = (unsigned)x_bytes; a.w_bytes = (unsigned)w_bytes; a.y_bytes = (unsigned)y_bytes;
    a.bias1 = nullptr; a.bias_ds = nullptr; a.o1 = nullptr; a.ods = nullptr; a.o1_bytes = 0; a.ods_bytes = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (wide) return W == 55 ? launch_w2<Gd4w, 0>(a, s) : launch_w2<Gr1w, 0>(a, s);
    if (pad == 0 && W == 56) return !fused ? launch_w2<Gd2, 0>(a, s) : inverse ? launch_w2<Gd2, 1, true>(a, s) : launch_w2<Gd2, 1, false>(a, s);
    if (pad == 1 && W == 55) return !fused ? launch_w2<Gd4, 0>(a, s) : inverse ? launch_w2<Gd4, 1, true>(a, s) : launch_w2<Gd4, 1, false>(a, s);
    if (pad == 0) return !fused ? launch_w2<Gr0, 0>(a, s) : inverse ? launch_w2<Gr0, 1, true>(a, s) : launch_w2<Gr0, 1, false>(a, s);
    return !fused ? launch_w2<Gr1, 0>(a, s) : inverse ? launch_w2<Gr1, 1, true>(a, s) : launch_w2<Gr1, 1, false>(a, s);
}

extern "C" int sc2_conv2x2_win_tail_supported(int H, int W, int Cin) {
    return sc2_conv2x2_win_supported(H, W, Cin, 256, 1) && H == 55 && W == 55 ? 1 : 0;
}

extern "C" int sc2_conv2x2_win_tail_fwd(const void *x, const void *w_stream, const float *bias1, const float *bias_ds, void *y,
                                        void *o1, void *ods, int N, int H, int W, int Cin, void *stream) {
    SC2_REQUIRE(x && w_stream && bias1 && bias_ds && o1 && ods, SC2_ERR_INVALID_ARG, "conv2x2_win_tail: null argument");
    SC2_REQUIRE(N > 0, SC2_ERR_INVALID_ARG, "conv2x2_win_tail: non-positive batch");
    SC2_REQUIRE(sc2_conv2x2_win_tail_supported(H, W, Cin), SC2_ERR_UNSUPPORTED,
                "conv2x2_win_tail: needs a 55 x 55 input (56 x 56 output), Cin %% 64 == 0 (got %d x %d, Cin %d)", H, W, Cin);
    const long long x_bytes = (long long)N * H * W * Cin * 2, y_bytes = (long long)N * 56 * 56 * 512;
    SC2_REQUIRE(x_bytes < 0x7FF00000LL && y_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv2x2_win_tail: tensor of %lld bytes exceeds 2 GB",
                x_bytes > y_bytes ? x_bytes : y_bytes);
    W2Args a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_stream);
    a.beta = nullptr;
    a.y = static_cast<uint16_t *>(y);
    a.bias1 = bias1; a.bias_ds = bias_ds;
    a.o1 = static_cast<uint16_t *>(o1);
    a.ods = static_cast<uint16_t *>(ods);
    a.N = N; a.H = H; a.Cin = Cin; a.OH = 56;
    a.W = W; a.OW = W + 1; a.nseg = 1;
    a.n_tiles = 0; a.tiles_per_img = 0; a.tiles_per_wg = 0;
    a.x_bytes = (unsigned)x_bytes; a.w_bytes = (unsigned)((Cin / 32 * 4 + 24) * 16384); a.y_bytes = y ? (unsigned)y_bytes : 0u;
    a.o1_bytes = (unsigned)((long long)N * 56 * 56 * 256); a.ods_bytes = (unsigned)((long long)N * 28 * 28 * 1024);
    return launch_w2<Gd4, 2>(a, static_cast<hipStream_t>(stream));
}
