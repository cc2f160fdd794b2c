// Second encoder stage of the FP / SHP / MSHP bottlenecks in ONE persistent launch (gfx950):
//     y = GDN1_48( Conv2d(96 -> 48, k5, s2, p2, bias=False)(x) )                (sc2bench/models/layer.py:479-481)
// for 112-pixel-wide inputs (the 224 x 224 path).  185 GFLOP per 256-image batch over 0.62 GB of input.
//
// Why its own kernel: K = 2400 against N = 48.  Every formulation that streams the weights per tile moves 230 KB of them
// for 112 output pixels (the LDS-patch tile kernel: 2.7 GB through L2 per batch, 0.36 ms in its K loop alone).  Here the
// weights never move again after the prologue:
//   * a workgroup is 4 waves - one per SIMD, each with the whole 512-entry register file - one workgroup per CU,
//     persistent; K is split over the waves BY (slab, tap): wave w owns taps ((w + cb) & 3) + 4q of channel slab cb -
//     18 or 19 of the 75 k-steps - and keeps their fragments in REGISTERS (54 resident fragments = 216 VGPRs, + the three of tap 24
//     for the one slab in which this wave has a seventh tap: round 6, they were fetched per slab before); no partner wave hides
//     latency, so the fragments of the next tap are read -- and the next slab's patch pieces issued -- between the MFMAs of the
//     current one (round 6: inside its seven groups of three MFMAs, not between two taps);
//   * a unit = two output rows of one image (112 pixels = exactly seven 16-row MFMA tiles x three channel tiles); its
//     input patch (7 rows x 116 columns) is staged per 32-channel slab by direct-to-LDS loads into a double buffer,
//     slab g + 1 in flight while slab g is multiplied (buffer descriptors: per-lane offsets are constants of the
//     kernel, the row / slab position is the scalar offset, out-of-image lanes read zeros);
//   * every wave accumulates partial sums for the WHOLE 112 x 48 tile; they are added up through LDS in two rounds of
//     four pixel tiles (48 KB), pixel tile 4 r + w summed by wave w, which then owns all 48 channels of its 16 pixels;
//   * GDN1(48) is therefore wave-local: |t| (bf16) to the wave's own rows of an LDS image (no barrier), norm = beta +
//     gamma |t| as six more MFMAs per owned tile (gamma fragments in LDS), y = t / norm, bf16 to a 10.5 KB output image
//     that is one contiguous block of the NHWC output.
// Units are claimed dynamically, two ahead, from one counter per XCD (claim c = local unit c + 2 x workgroups of the
// XCD; an XCD makes exactly as many claims as it has units, and the last one re-arms its counter).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "sc2_common.h"

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(f32x2_t v) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
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

__device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// Round 4: EVERY LDS access of the unit loop is inline asm and every barrier a raw s_barrier.  hipcc knows that a direct-to-LDS
// load writes LDS: in front of any LDS access it can see -- and of every __syncthreads() -- it drains vmcnt(0).  The next slab's
// (or unit's) pieces are in flight all the time, so each such access waited for them (a loaded HBM round trip: the reduction
// rounds took 4.5 - 7 k of a unit's 22 - 24 k cycles, tools/enc2_stamps.py) and the counted `vmcnt(3)` at a unit's first slab was
// void.  Results of asm reads are consumed behind `lds_wait()` = s_waitcnt lgkmcnt(0) + a scheduling barrier.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write16(uint32_t addr, uint4 v) {
    const u32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(q) : "memory");
}
__device__ __forceinline__ void lds_write8(uint32_t addr, uint2 v) {
    const u32x2_t q = {v.x, v.y};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(q) : "memory");
}
__device__ __forceinline__ void lds_write4(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wg_barrier() {   // this wave's LDS writes are done, then the workgroup meets (no vmcnt drain)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

#ifndef SC2_ENC2_SCHED
#define SC2_ENC2_SCHED 1   // 1 (round 6): the slab loop with the patch pieces, the per-unit piece offsets and the next tap's fragment reads placed
#endif                     //    BESIDE the MFMAs of a tap instead of between two taps; 0: the round-5 loop (A/B: tools/build_variant.sh)
#ifndef SC2_ENC2_W6
#define SC2_ENC2_W6 1   // 1 (round 6): the seventh tap's fragments are resident; 0: fetched per slab
#endif
#ifndef SC2_ENC2_DBG
#define SC2_ENC2_DBG 0   // timing experiment only (WRONG results): 8 = every unit fetches the rows of unit 0, i.e. every patch piece hits L2
#endif
#ifndef SC2_ENC2_STAMPS
#define SC2_ENC2_STAMPS 0   // 1: diagnostic build that records s_memtime at the phase boundaries (tools/enc2_stamps.py)
#endif
#if SC2_ENC2_STAMPS
// (mode 2: 40 slots per unit -- the phase boundaries in slots 30 .. 39, slot 10 cb + t inside slab cb: t = 0 first fragments issued, 1 + q behind tap q)
constexpr int N_STAMP = SC2_ENC2_STAMPS == 2 ? 40 : 12;
#define STAMP_AT(k)                                                                                             \
    do {                                                                                                        \
        if (p.stamps && lane == 0 && blockIdx.x < 8 && g_units < 16)                                            \
            p.stamps[((blockIdx.x * 4 + wave) * 16 + g_units) * N_STAMP + (k)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#define STAMP(k) STAMP_AT((SC2_ENC2_STAMPS == 2 ? 30 : 0) + (k))
#if SC2_ENC2_STAMPS == 2
#define FSTAMP(cb, t) STAMP_AT(10 * (cb) + (t))
#else
#define FSTAMP(cb, t)
#endif
#else
constexpr int N_STAMP = 12;
#define STAMP(k)
#define FSTAMP(cb, t)
#endif

struct Enc2Args {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, 112, 96]
    const uint16_t *__restrict__ w;      // bf16 fragment-major, slab-major K: [k-step = cb*25 + tap][3][64][8]
    const uint16_t *__restrict__ g;      // bf16 fragment-major gamma [3][2][64][8] (K 48 zero-padded to 64)
    const float *__restrict__ beta;      // f32 [48]
    uint16_t *__restrict__ y;            // bf16 NHWC [N, OH, 56, 48]
    uint16_t *__restrict__ t_out;        // EMIT instantiations: the conv output in front of the GDN (bf16, laid out like y)
    int N, H, OH, n_units, units_per_img;
    int W, OWT, n_seg;    // SEG instantiations (any width): input width, output width, 56-column segments per output row
    unsigned *unit_ctr;   // eight counters: one per XCD
    unsigned long long *stamps;   // diagnostic build only (SC2_ENC2_STAMPS)
};

constexpr int W_IN = 112, OW = 56, CIN = 96, COUT = 48, J = OW + 2, NTAP = 25, NCB = 3;
constexpr int MT = 7, NT = 3;
constexpr int N_CHUNKS = 2 * 7 * J * 4;                 // 16-byte chunks of one slab's patch (3248)
constexpr int N_PIECES = (N_CHUNKS + 63) / 64;          // 1 KB direct-to-LDS pieces (51)
constexpr int PATCH_STRIDE = N_PIECES * 1024;           // 52 224
constexpr int NW = 4, NQ = 6;                           // waves; resident taps per wave and slab
constexpr int PIECES_PER_WAVE = (N_PIECES + NW - 1) / NW;   // 13
constexpr int RED_OFF = 2 * PATCH_STRIDE;               // reduction rounds: [4 waves][12 blocks][64 lanes][16 B] = 48 KB
constexpr int RED_BYTES = NW * 12 * 1024;
constexpr int TIMG_OFF = RED_OFF;                       // |t| image [112][128 B] (chunks 6, 7 stay zero), over the dead rounds
constexpr int OIMG_OFF = RED_OFF + 112 * 128;           // output image [112][96 B]
constexpr int XIMG_OFF = OIMG_OFF + 112 * 96;           // EMIT: image of the conv output t [112][96 B]
constexpr int GAM_OFF = RED_OFF + RED_BYTES;            // gamma fragments [3][2][64] x 16 B, then beta [48] f32: loaded once
constexpr int BETA_OFF = GAM_OFF + 6 * 1024;
constexpr int LDS_BYTES = BETA_OFF + COUT * 4;
static_assert(XIMG_OFF + 112 * 96 <= GAM_OFF, "images fit the reduction area");
static_assert(LDS_BYTES + 64 <= 160 * 1024, "one workgroup per CU");
static_assert(NW * 9 * 1024 <= PATCH_STRIDE, "pixel tiles 4 .. 6 of the four partial sums fit a dead patch buffer");

// SEG (round 4): any input width.  An output row is cut into segments of OW = 56 pixels (a unit = two output rows of ONE
// segment), the patch of a segment starts 2 ow0 input columns to the right, and which of its columns exist is decided per
// unit instead of once per workgroup; everything behind the patch fill -- tap offsets, fragment addresses, reduction, GDN1 --
// is segment-local and unchanged.  !SEG is the 112-pixel-wide geometry of the 224 x 224 operating point (one segment).
// EMIT (round 5, training): the conv output t in front of the GDN leaves too (bf16, the tensor the GDN's backward needs) -- the training
// forward then is this one launch instead of conv (0.49 ms) + GDN (0.06): three more 16-byte stores per thread and unit.
template <bool INVERSE, bool SEG, bool EMIT = false>
__global__ __launch_bounds__(256, 1) void conv2_gdn48_kernel(const Enc2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int next_slot;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const uint32_t next_slot_addr = (uint32_t)(uintptr_t)(lds_ptr_t)&next_slot;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 4);
    const int frow = lane & 15, fq = lane >> 4;
    const int H = p.H;
    const int W = SEG ? p.W : W_IN;            // input width (compile-time unless SEG)
    const int OWT = SEG ? p.OWT : OW;          // output width

    // ---- resident weight fragments: taps ((wave + cb) & 3) + 4 q, q < 6, of every slab (54 fragments = 216 VGPRs)
    const uint4 *wfrag = reinterpret_cast<const uint4 *>(p.w) + lane;   // [(kstep * 3 + j) * 64]
    uint4 wreg[NCB][NQ][NT];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int tap = ((wave + cb) & 3) + 4 * q;   // < 24
#pragma unroll
            for (int j = 0; j < NT; ++j) wreg[cb][q][j] = wfrag[((cb * NTAP + tap) * NT + j) * 64];
        }
    // Round 6: tap 24 is the seventh tap of exactly one (wave, slab) pair per wave -- slab (4 - wave) & 3; wave 1 has none -- so its three fragments
    // stay in registers too (the 12 the per-slab copy occupied anyway) instead of being fetched in every slab of every unit.  What that removes
    // is 9 loads per unit and the counted wait in front of the seventh tap: - 1.5 % on the launch (profiles/r06_enc2_taps.txt).
    constexpr bool W6_RESIDENT = SC2_ENC2_W6 != 0;   // (every instantiation: 475 registers in the SEG ones, 506 / 510 in the others)
    [[maybe_unused]] uint4 w6res[NT];
    if constexpr (W6_RESIDENT) {
        const int cb6 = (4 - wave) & 3;
#pragma unroll
        for (int j = 0; j < NT; ++j) w6res[j] = wfrag[(((cb6 < NCB ? cb6 : 0) * NTAP + (NTAP - 1)) * NT + j) * 64];
    }
    if (tid < 384 - 256) reinterpret_cast<uint4 *>(smem + GAM_OFF)[tid + 256] = reinterpret_cast<const uint4 *>(p.g)[tid + 256];
    reinterpret_cast<uint4 *>(smem + GAM_OFF)[tid] = reinterpret_cast<const uint4 *>(p.g)[tid];
    if (tid < COUT) reinterpret_cast<float *>(smem + BETA_OFF)[tid] = p.beta[tid];

    // ---- patch fill: piece pc = wave + 4 k of a slab = LDS chunks [64 pc, 64 pc + 64); chunk P <-> (plane, row r, half
    //      column j, physical chunk): source = input row 2 oh0 - 2 + r, column 2 j + plane - 2, logical chunk = phys ^ swz
    // (SEG: the descriptor base also sits two COLUMNS to the left, offsets are relative to column 2 ow0 - 2 and non-negative,
    //  and the column test is made per unit from pcol: issue_piece)
    uint32_t pv[PIECES_PER_WAVE];     // byte offset inside the image relative to row 2 oh0 - 2; low bits: r
    [[maybe_unused]] int pcol[SEG ? PIECES_PER_WAVE : 1];   // SEG: the chunk's input column relative to 2 ow0 (-2 .. 115)
#pragma unroll
    for (int k = 0; k < PIECES_PER_WAVE; ++k) {
        const int P = (wave + NW * k) * 64 + lane;
        const int cphys = P & 3, t = P >> 2;
        const int t2 = t / J, j = t - t2 * J;
        const int plane = t2 / 7, r = t2 - plane * 7;
        const int chunk = cphys ^ ((j >> 1) & 3);
        const int iw = 2 * j + plane - 2;
        if constexpr (SEG) {
            pcol[k] = iw;
            pv[k] = P < N_CHUNKS ? (uint32_t)(((r * W + iw + 2) * CIN + chunk * 8) * 2) | (uint32_t)r : 0x80000000u;
        } else {
            const bool ok = (P < N_CHUNKS) & ((unsigned)iw < (unsigned)W_IN);
            pv[k] = ok ? (uint32_t)(((r * W_IN + iw) * CIN + chunk * 8) * 2) | (uint32_t)r : 0x80000000u;
        }
    }
    // direct-to-LDS pieces of one slab: the descriptor / scalar offset / destination are set up once (patch_setup), the
    // pieces are then issued one or two at a time BETWEEN the taps of the slab being multiplied (a burst of 13 at the
    // top of a slab cost ~2 000 cycles of this wave's only instruction stream)
    struct PatchJob { buf_rsrc_t rs; uint32_t soff; unsigned char *dst; int row0; int col0; bool live; };
    // Units are XCD-local: workgroup b runs on XCD b & 7 (round-robin dispatch) and only takes units of the images
    // im = xcd (mod 8), in order - neighbouring row pairs of an image (3 of their 7 input rows are shared) and the three
    // slab passes over the same 128-byte lines then meet in ONE 4 MB L2 instead of eight.
    const int xcd = blockIdx.x & 7;
    const int wg_l = blockIdx.x >> 3;                                   // index of this workgroup on its XCD
    const int wgs_x = ((int)gridDim.x - xcd + 7) >> 3;                  // workgroups on this XCD
    const int n_local = xcd < p.N ? ((p.N - xcd + 7) >> 3) * p.units_per_img : 0;   // units of this XCD
    unsigned *const my_ctr = p.unit_ctr + xcd;
    auto patch_setup = [&](int unit, int cb, int buf) {
        PatchJob jb;
        jb.live = unit < n_local;
        const int im_l = jb.live && !(SC2_ENC2_DBG & 8) ? unit / p.units_per_img : 0;   // (DBG 8: every unit fetches the rows of unit 0 -- L2 hits)
        const int im = xcd + 8 * im_l;
        const int u_in = jb.live && !(SC2_ENC2_DBG & 8) ? unit - im_l * p.units_per_img : 0;
        const int rp = SEG ? u_in / p.n_seg : u_in, seg = SEG ? u_in - rp * p.n_seg : 0;
        const int oh0 = rp * 2, ow0 = seg * OW;
        // descriptor base two rows above the image (SEG: and two columns to its left): offsets are then non-negative
        jb.rs = make_rsrc(p.x + (((long long)im * H - 2) * W - (SEG ? 2 : 0)) * CIN, (uint32_t)((H + 2) * W + (SEG ? 2 : 0)) * CIN * 2);
        jb.soff = (uint32_t)(((2 * oh0) * W + 2 * ow0) * CIN + cb * 32) * 2u;
        jb.dst = smem + buf * PATCH_STRIDE;
        jb.row0 = 2 * oh0 - 2;
        jb.col0 = 2 * ow0;
        return jb;
    };
    // (branch-free: this wave is alone on its SIMD and a scalar branch costs it tens of cycles -- the unit loop held 109 of them.
    //  A piece index past the end (k = 12 on the last wave) re-issues the wave's FIRST piece: same bytes to the same place.)
    auto issue_piece = [&](const PatchJob &jb, int k) {
        const bool real = k < PIECES_PER_WAVE - 1 || wave + NW * k < N_PIECES;   // (k < 12: compile-time true)
        const int pc = real ? wave + NW * k : wave;
        const uint32_t pvk = real ? pv[k] : pv[0];
        const int ih = jb.row0 + (int)(pvk & 7u);   // patch row r rides in the offset's free low bits
        bool ok = jb.live & ((unsigned)ih < (unsigned)H);
        if constexpr (SEG) ok = ok & ((int)pvk >= 0) & ((unsigned)(jb.col0 + (real ? pcol[k] : pcol[0])) < (unsigned)W);
        const uint32_t vo = ok ? (pvk & ~15u) : 0x80000000u;
        buf_load_lds16(jb.rs, (lds_ptr_t)(jb.dst + pc * 1024), vo, jb.soff);
    };
    auto issue_patch = [&](int unit, int cb, int buf) {
        const PatchJob jb = patch_setup(unit, cb, buf);
#pragma unroll
        for (int k = 0; k < PIECES_PER_WAVE; ++k) issue_piece(jb, k);
    };
    constexpr bool SCHED = SC2_ENC2_SCHED && !SEG;   // (the SEG instantiations hold 465 registers in the round-5 loop: this one spills there)
    // Round 6: what a piece's offset depends on beside the lane -- which patch rows (SEG: and columns) exist -- is a property of the UNIT, the
    // same for its three slabs.  pvo[k] holds it for the unit whose patch is being fetched: computed once per unit (in slab 1, entry by entry
    // behind the piece that used the old value, for the unit slab 2 starts fetching), so that issuing a piece is an LDS address and a load.
    // What the per-tap stamps of the round-5 loop showed (`tools/enc2_stamps.py` on a -DSC2_ENC2_STAMPS=2 build, profiles/r06_enc2_taps.txt):
    // 515 cycles per tap in slab 1, 650 - 950 in slabs 0 and 2, against 336 of MFMAs.  ~180 of them were the instructions BETWEEN two taps (piece
    // offsets, piece issue, the wait for the last fragment read): those now ride beside the MFMAs, - 2 % on the launch.  The rest is not
    // instruction count: slab 1 fetches slab 2 of the same rows, whose lines slabs 0 and 1 have already pulled into L2; slabs 0 and 2 fetch
    // lines that come from HBM, and a wave whose piece waits for a place in the CU's vector-memory queue issues no MFMA either (one wave per SIMD).
    // With every row in L2 (`-DSC2_ENC2_DBG=8`: each unit fetches the rows of unit 0) the launch takes 0.227 instead of 0.286 ms on the same box.
    // Tried against that and NOT kept: an L2 prefetch of the unit after next (one dword per line, direct-to-LDS into a junk area: + 12 %, the
    // prefetch loads stall the wave just the same and two units ahead do not fit 4 MB of L2 beside 32 workgroups' patches), the pieces of a slab
    // issued hit / miss alternately or misses last (+ 0.7 %), the first slab's pieces one tap late (+ 1 %), the images walked backwards (the
    // producer's latest output first: no difference), the encoder in slices of 64 images that fit the memory-side cache (slower).
    uint32_t pvo[PIECES_PER_WAVE];
    auto piece_vo = [&](const PatchJob &jb, int k) {
        const bool real = k < PIECES_PER_WAVE - 1 || wave + NW * k < N_PIECES;
        const uint32_t pvk = real ? pv[k] : pv[0];
        const int ih = jb.row0 + (int)(pvk & 7u);
        bool ok = jb.live & ((unsigned)ih < (unsigned)H);
        if constexpr (SEG) ok = ok & ((int)pvk >= 0) & ((unsigned)(jb.col0 + (real ? pcol[k] : pcol[0])) < (unsigned)W);
        return ok ? (pvk & ~15u) : 0x80000000u;
    };
    auto issue_piece_vo = [&](const PatchJob &jb, int k) {
        const bool real = k < PIECES_PER_WAVE - 1 || wave + NW * k < N_PIECES;
        const int pc = real ? wave + NW * k : wave;
        buf_load_lds16(jb.rs, (lds_ptr_t)(jb.dst + pc * 1024), pvo[k], jb.soff);
    };
    // fragment address of (pixel tile i, this lane) at tap offset tap_off / half-column shift d inside patch buffer pb:
    // the pixel part is a constant of the lane (14 registers); per read only the shift's swizzle is rebuilt - the ~10
    // instructions of the full form per read did not fit beside the MFMAs of this wave's only instruction stream
    uint32_t fa_base[MT];
    int fa_col[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = i * 16 + frow;
        const int orow = pix >= OW ? 1 : 0;
        fa_col[i] = pix - orow * OW;
        fa_base[i] = (uint32_t)(((2 * orow) * J + fa_col[i]) * 64);
    }
    auto frag_addr = [&](uint32_t pb_tap, int i, int d, int fqo) {   // pb_tap = patch buffer + tap offset (scalar)
        return pb_tap + fa_base[i] + (uint32_t)((fqo ^ (((fa_col[i] + d) >> 1) & 3)) << 4);   // fqo: opaque copy of fq
    };

    int unit = wg_l;
    int next_unit = unit + wgs_x;
    issue_patch(unit, 0, 0);
    if constexpr (SCHED) {
        const PatchJob jb0 = patch_setup(unit, 0, 0);
#pragma unroll
        for (int k = 0; k < PIECES_PER_WAVE; ++k) pvo[k] = piece_vo(jb0, k);
    }
    int g = 0;   // global slab counter of this workgroup: slab g lives in patch buffer g & 1
    [[maybe_unused]] int g_units = 0;

    while (unit < n_local) {
        const int im_l = unit / p.units_per_img;
        const int im = xcd + 8 * im_l;
        const int u_in = unit - im_l * p.units_per_img;
        const int rp = SEG ? u_in / p.n_seg : u_in, seg = SEG ? u_in - rp * p.n_seg : 0;
        const int oh0 = rp * 2;
        [[maybe_unused]] const int ow0 = seg * OW;
        const int n_rows = p.OH - oh0 >= 2 ? 2 : 1;
        [[maybe_unused]] const int n_cols = OWT - ow0 >= OW ? OW : OWT - ow0;   // valid output columns of this segment
        STAMP(0);
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        unsigned claimed = 0;

#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            // slab g has landed (this wave's pieces), then everybody's; the barrier also says every wave is done reading
            // slab g - 1, whose buffer slab g + 1 now overwrites.  (First slab of a unit: only the three output stores of
            // the previous unit were issued after this slab's loads, so vmcnt(3) does not wait for their acknowledgements.)
            if (cb == 0 && g != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EMIT ? 6 : 3) : "memory");   // (EMIT: six output stores)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            STAMP(1 + 2 * cb);
            // Claim of the unit after next: ISSUED behind the first slab's barrier, READ behind the second slab's `vmcnt(0)` above
            // (which the wave executes anyway), so its round trip -- 0.3 - 1 us with 32 workgroups on an XCD's counter -- runs
            // beside a whole slab of MFMAs instead of stalling wave 0 (and, at the next barrier, everybody).  The destination
            // register is written when the atomic RETURNS; hipcc does not know that and once copied such a register early (stale
            // claims, an endless unit loop): tools/audit_asm_atomic.py checks in the built ISA that its first reader is the add
            // below, behind the wait (tests/test_abi.py runs it).
            if (cb == 0 && tid == 0) {
                const unsigned one = 1u;
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(my_ctr), "v"(one) : "memory");
            }
            if (cb == 1 && tid == 0) {
                asm volatile("" : "+v"(claimed)::"memory");
                lds_write4(next_slot_addr, claimed + 2u * (unsigned)wgs_x);
                if (claimed == (unsigned)(n_local - 1)) *my_ctr = 0u;   // this XCD's last claim re-arms its counter
            }
            // (SC2_ENC2_W6 = 0, round 5: the wave with a seventh tap in this slab fetches that fragment now -- L2; older than the patch loads below)
            [[maybe_unused]] const int tap6 = ((wave + cb) & 3) + 4 * NQ;
            // (asm loads, waited for with a counted vmcnt in front of the seventh tap: as compiler-tracked loads their wait was
            //  vmcnt(0) = for the 12 - 13 patch pieces this wave issues behind them, i.e. the wave with seven taps -- the slowest
            //  of the slab already -- also waited for the next slab's or unit's patch)
            u32x4_t w6[NT];
            if constexpr (W6_RESIDENT) {
#pragma unroll
                for (int j = 0; j < NT; ++j) w6[j] = u32x4_t{w6res[j].x, w6res[j].y, w6res[j].z, w6res[j].w};
            } else {   // (every wave fetches: the three without a seventh tap load tap 24 again and ignore it -- no branch, and the same
                //  number of vector-memory operations in flight on every wave)
                int ln = lane;
                asm volatile("" : "+v"(ln));   // (address rebuilt here, not carried across the unit)
                const uint4 *w6p = reinterpret_cast<const uint4 *>(p.w) + ((cb * NTAP + (tap6 < NTAP ? tap6 : NTAP - 1)) * NT) * 64 + ln;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    asm volatile("global_load_dwordx4 %0, %1, off ; wfrag" : "=&v"(w6[j]) : "v"(w6p + j * 64) : "memory");
            }
            // (the slab behind the unit's last one is the NEXT unit's first: same buffer rotation, and since round 4 issued behind
            //  the taps like every other slab -- in the reduction rounds, where it used to be issued, the 13 pieces cost more than
            //  among MFMAs and the rounds were the longest phase of a unit: 7.4 k of 23.8 k cycles, tools/enc2_stamps.py)
            const PatchJob jb = cb + 1 < NCB ? patch_setup(unit, cb + 1, (g + 1) & 1) : patch_setup(next_unit, 0, (g + 1) & 1);
            const uint32_t pb = lds_base + (uint32_t)((g & 1) * PATCH_STRIDE);
            // one wave per SIMD: the fragments of tap q + 1 are read between the MFMAs of tap q
            uint4 av[2][MT];
            {
                const int tap = (wave + cb) & 3;
                const int kh = tap / 5, kw = tap - kh * 5;
                int fqo = fq;
                asm volatile("" : "+v"(fqo));   // (keeps the 7 x 25 complete addresses from being hoisted and spilled)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    av[0][i] = lds_read16(frag_addr(pb + (uint32_t)((((kw & 1) * 7 + kh) * J + (kw >> 1)) * 64), i, kw >> 1, fqo));
            }
            FSTAMP(cb, 0);
            if constexpr (SCHED) {
            // (for slab 1: the unit whose first slab the NEXT slab starts fetching -- its piece offsets replace this unit's one by one)
            [[maybe_unused]] const PatchJob jbn = patch_setup(next_unit, 0, 0);
#pragma unroll
            for (int q = 0; q <= NQ; ++q) {
                const int tap = ((wave + cb) & 3) + 4 * q;
                // (seventh tap: its fragments are older than the >= 12 patch pieces issued beside taps 0 .. 4.  EVERY wave waits,
                //  also the three that ignore what they fetched: a load that lands in a register the compiler has given to
                //  something else is the hazard tools/audit_vmcnt.py --copies looks for; the marker tells it they have landed)
                if (q == NQ && !W6_RESIDENT) asm volatile("s_waitcnt vmcnt(12) ; wfrag-landed" ::: "memory");
                if (tap < NTAP) {   // wave-uniform (q < NQ: always)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tap q's fragments
                    __builtin_amdgcn_sched_barrier(0);
                    const int ntap = tap + 4 < NTAP ? tap + 4 : tap;   // (past the last tap: this tap's fragments again, unused)
                    const int nkh = ntap / 5, nkw = ntap - nkh * 5;
                    const int noff = (((nkw & 1) * 7 + nkh) * J + (nkw >> 1)) * 64;
                    int fqo = fq;
                    asm volatile("" : "+v"(fqo));
                    // A tap = seven groups of three MFMAs (48 cycles of the matrix pipe), and everything else the wave has to do rides in
                    // them, a few instructions per group: the next tap's fragment reads in groups 0 .. 3 (two each: the last one then has
                    // nine MFMAs to land behind, not two), a patch piece in groups 1, 3, 5 of taps 0 .. 3 (the thirteenth in tap 4), and in
                    // slab 1 the next unit's offset for that piece one group later.
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        if (q < NQ) {
                            if (i < 3) {
                                av[(q + 1) & 1][2 * i] = lds_read16(frag_addr(pb + (uint32_t)noff, 2 * i, nkw >> 1, fqo));
                                av[(q + 1) & 1][2 * i + 1] = lds_read16(frag_addr(pb + (uint32_t)noff, 2 * i + 1, nkw >> 1, fqo));
                            } else if (i == 3) {
                                av[(q + 1) & 1][6] = lds_read16(frag_addr(pb + (uint32_t)noff, 6, nkw >> 1, fqo));
                            }
                        }
                        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[q & 1][i]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bf16x8_t wv = q < NQ ? __builtin_bit_cast(bf16x8_t, wreg[cb][q < NQ ? q : 0][j])
                                                       : __builtin_bit_cast(bf16x8_t, w6[j]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, af, acc[i][j], 0, 0, 0);
                        }
                        const int k_issue = q < 4 ? ((i & 1) && i < 6 ? 3 * q + (i >> 1) : -1) : (q == 4 && i == 1 ? 12 : -1);
                        const int k_next = q < 4 ? (!(i & 1) && i >= 2 ? 3 * q + (i >> 1) - 1 : -1) : (q == 4 && i == 2 ? 12 : -1);
                        if (k_issue >= 0) issue_piece_vo(jb, k_issue);
                        if (cb == 1 && k_next >= 0) pvo[k_next] = piece_vo(jbn, k_next);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                FSTAMP(cb, 1 + q);
            }
            } else {
#pragma unroll
            for (int q = 0; q <= NQ; ++q) {
                const int tap = ((wave + cb) & 3) + 4 * q;
                // (seventh tap: its fragments are older than the >= 12 patch pieces issued behind taps 0 .. 4.  EVERY wave waits,
                //  also the three that ignore what they fetched: a load that lands in a register the compiler has given to
                //  something else is the hazard tools/audit_vmcnt.py --copies looks for; the marker tells it they have landed)
                if (q == NQ && !W6_RESIDENT) asm volatile("s_waitcnt vmcnt(12) ; wfrag-landed" ::: "memory");
                if (tap < NTAP) {   // wave-uniform (q < NQ: always)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tap q's fragments
                    __builtin_amdgcn_sched_barrier(0);
                    const int ntap = tap + 4 < NTAP ? tap + 4 : tap;   // (past the last tap: this tap's fragments again, unused)
                    const int nkh = ntap / 5, nkw = ntap - nkh * 5;
                    const int noff = (((nkw & 1) * 7 + nkh) * J + (nkw >> 1)) * 64;
                    int fqo = fq;
                    asm volatile("" : "+v"(fqo));

#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        if (q < NQ) av[(q + 1) & 1][i] = lds_read16(frag_addr(pb + (uint32_t)noff, i, nkw >> 1, fqo));
                        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[q & 1][i]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bf16x8_t wv = q < NQ ? __builtin_bit_cast(bf16x8_t, wreg[cb][q < NQ ? q : 0][j])
                                                       : __builtin_bit_cast(bf16x8_t, w6[j]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, af, acc[i][j], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // three pieces of the next slab behind each of the first taps (13 pieces over taps 0 .. 4): spread, but
                // early enough to have landed when the slab ends
                {
                    if (q < 4) {
                        issue_piece(jb, 3 * q);
                        issue_piece(jb, 3 * q + 1);
                        issue_piece(jb, 3 * q + 2);
                    } else if (q == 4) {
                        issue_piece(jb, 12);
                    }
                }
                FSTAMP(cb, 1 + q);
            }
            }
            STAMP(2 + 2 * cb);
            ++g;
        }
        // ---------------------------------------------------------------- sum of the four partial tiles, ONE round
        // Every wave parks its 21 partial tiles: pixel tiles 0 .. 3 in the round area (48 KB), 4 .. 6 in the patch buffer of the
        // unit's LAST slab, which is dead until the next unit's second slab is issued (36 of its 52 KB) -- one write / read round
        // and three barriers instead of two rounds and five (round 4).  Same summation order as before (wave 0 + 1 + 2 + 3).
        f32x4_t own[2][NT];   // owned pixel tiles 4 r + wave, all three channel tiles
        wg_barrier();   // everybody is done reading the last slab
        const uint32_t red_lo_a = lds_base + (uint32_t)RED_OFF, red_hi_a = lds_base + (uint32_t)(((g - 1) & 1) * PATCH_STRIDE);
        int ln_r = lane;
        asm volatile("" : "+v"(ln_r));   // (slot addresses formed here, not carried across the slabs)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                lds_write16(i < 4 ? red_lo_a + (uint32_t)(((wave * 12 + i * 3 + j) * 64 + ln_r) * 16)
                                  : red_hi_a + (uint32_t)(((wave * 9 + (i - 4) * 3 + j) * 64 + ln_r) * 16),
                            __builtin_bit_cast(uint4, acc[i][j]));
        wg_barrier();
        // six batches of four partial tiles (one per wave), two batches in flight: 32 registers (all 24 reads at once spilled
        // resident weight fragments)
        auto red_addr = [&](int b, int w2) {
            const int r = b / NT, j = b - r * NT;
            return r == 0 ? red_lo_a + (uint32_t)(((w2 * 12 + wave * 3 + j) * 64 + ln_r) * 16)
                          : red_hi_a + (uint32_t)(((w2 * 9 + (4 + wave < MT ? wave : 0) * 3 + j) * 64 + ln_r) * 16);   // (wave 3 owns no
        };                                                                                                     // second tile: reads wave 0's)
        uint4 pa[2][NW];
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) pa[0][w2] = lds_read16(red_addr(0, w2));
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) pa[1][w2] = lds_read16(red_addr(1, w2));
#pragma unroll
        for (int b = 0; b < 2 * NT; ++b) {
            if (b + 1 < 2 * NT) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");   // batch b landed, batch b + 1 may be in flight
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (4 * (b / NT) + wave < MT) {
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) sum += __builtin_bit_cast(f32x4_t, pa[b & 1][w2]);
            }
            own[b / NT][b % NT] = sum;
            __builtin_amdgcn_sched_barrier(0);
            if (b + 2 < 2 * NT) {
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) pa[b & 1][w2] = lds_read16(red_addr(b + 2, w2));
            }
        }
        wg_barrier();   // the round is dead: its area becomes the |t| image and the output image
        STAMP(7);
        // ---------------------------------------------------------------- GDN1(48) on the owned pixel tiles (wave-private rows)
        const uint32_t timg = lds_base + (uint32_t)TIMG_OFF, oimg = lds_base + (uint32_t)OIMG_OFF;
        [[maybe_unused]] const uint32_t ximg = lds_base + (uint32_t)XIMG_OFF;
        {
            // both owned pixel tiles side by side (their LDS round trips and MFMA chains overlap); the rows are this wave's
            // own, its LDS operations complete in order: no barrier
            const bool has1 = 4 + wave < MT;   // wave-uniform: wave 3 owns one tile only
            // (lane coordinates from the opaque copy made in this unit: hoisted out of the unit loop, the image addresses derived
            //  from them were spilled, and a scratch reload waits vmcnt(0))
            const int frow = ln_r & 15, fq = ln_r >> 4;
            int px[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) px[r] = (4 * r + wave) * 16 + frow;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && !has1) continue;
                if (fq < 2) lds_write16(timg + (uint32_t)(px[r] * 128 + (((6 + fq) ^ (px[r] & 7)) << 4)), make_uint4(0u, 0u, 0u, 0u));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int c = j * 2 + (fq >> 1);
                    uint2 h;
                    h.x = pack2(f32x2_t{own[r][j][0], own[r][j][1]}) & 0x7FFF7FFFu;
                    h.y = pack2(f32x2_t{own[r][j][2], own[r][j][3]}) & 0x7FFF7FFFu;
                    lds_write8(timg + (uint32_t)(px[r] * 128 + ((c ^ (px[r] & 7)) << 4) + (fq & 1) * 8), h);
                }
            }
            uint4 gv[NT][2];
            float4 beta4[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                beta4[j] = __builtin_bit_cast(float4, lds_read16(lds_base + (uint32_t)(BETA_OFF + (j * 16 + fq * 4) * 4)));
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    gv[j][ks] = lds_read16(lds_base + (uint32_t)(GAM_OFF + ((j * 2 + ks) * 64 + ln_r) * 16));
            }
            uint4 xv[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xv[r][ks] = lds_read16(timg + (uint32_t)(px[r] * 128 + (((ks * 4 + fq) ^ (px[r] & 7)) << 4)));
            lds_wait();   // (a wave's LDS operations complete in order: its own |t| rows are written before they are read)
            f32x4_t nrm[2][NT];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    nrm[r][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        nrm[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, gv[j][ks]),
                                                                            __builtin_bit_cast(bf16x8_t, xv[r][ks]), nrm[r][j], 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && !has1) continue;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f32x2_t n01 = f32x2_t{beta4[j].x, beta4[j].y} + f32x2_t{nrm[r][j][0], nrm[r][j][1]};
                    const f32x2_t n23 = f32x2_t{beta4[j].z, beta4[j].w} + f32x2_t{nrm[r][j][2], nrm[r][j][3]};
                    const f32x2_t t01 = {own[r][j][0], own[r][j][1]}, t23 = {own[r][j][2], own[r][j][3]};
                    uint2 o;
                    if (INVERSE) {
                        o.x = pack2(t01 * n01);
                        o.y = pack2(t23 * n23);
                    } else {
                        o.x = pack2(t01 * f32x2_t{__builtin_amdgcn_rcpf(n01[0]), __builtin_amdgcn_rcpf(n01[1])});
                        o.y = pack2(t23 * f32x2_t{__builtin_amdgcn_rcpf(n23[0]), __builtin_amdgcn_rcpf(n23[1])});
                    }
                    lds_write8(oimg + (uint32_t)(px[r] * (COUT * 2) + (j * 16 + fq * 4) * 2), o);
                    if constexpr (EMIT) {
                        uint2 tx;
                        tx.x = pack2(t01);
                        tx.y = pack2(t23);
                        lds_write8(ximg + (uint32_t)(px[r] * (COUT * 2) + (j * 16 + fq * 4) * 2), tx);
                    }
                }
            }
        }
        wg_barrier();
        STAMP(8);
        // ---------------------------------------------------------------- stream the unit out (one contiguous block of y)
        if constexpr (!SEG) {
            uint4 *yo = reinterpret_cast<uint4 *>(p.y + ((long long)(im * p.OH + oh0) * OW) * COUT);
            const unsigned n_out = (unsigned)(n_rows * OW * (COUT / 8));   // 672 or 336 chunks
            uint4 ov[3];
            unsigned oq[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned tid_r = (unsigned)(wave * 64 + ln_r);
                const unsigned q0 = tid_r + 256 * k;
                oq[k] = q0 < n_out ? q0 : tid_r;   // past the end: the thread's first chunk again (same data)
                ov[k] = lds_read16(oimg + oq[k] * 16u);
            }
            [[maybe_unused]] uint4 xv3[3];
            if constexpr (EMIT) {
#pragma unroll
                for (int k = 0; k < 3; ++k) xv3[k] = lds_read16(ximg + oq[k] * 16u);
            }
            lds_wait();
#pragma unroll
            for (int k = 0; k < 3; ++k) yo[oq[k]] = ov[k];
            if constexpr (EMIT) {
                uint4 *xo = reinterpret_cast<uint4 *>(p.t_out + ((long long)(im * p.OH + oh0) * OW) * COUT);
#pragma unroll
                for (int k = 0; k < 3; ++k) xo[oq[k]] = xv3[k];
            }
        } else {
            // the unit's two output rows x n_cols pixels: runs of n_cols * 96 bytes at (oh0 + row, ow0); a chunk outside them is
            // replaced by one of row 0 that always exists (same data, same address: still three stores per thread)
            uint4 *yo = reinterpret_cast<uint4 *>(p.y + (((long long)im * p.OH + oh0) * OWT + ow0) * COUT);
            const unsigned tid_r = (unsigned)(wave * 64 + ln_r);
            const unsigned q_safe = tid_r % (unsigned)(6 * n_cols);
            uint4 ov[3];
            [[maybe_unused]] uint4 xv3[3];
            unsigned oa[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned q0 = tid_r + 256 * k;
                const unsigned px0 = (q0 * 10923u) >> 16;                         // q0 / 6 for q0 < 768
                const unsigned row0 = px0 >= (unsigned)OW ? 1u : 0u, col0 = px0 - row0 * OW;
                const bool ok = (q0 < (unsigned)(2 * OW * 6)) & (row0 < (unsigned)n_rows) & (col0 < (unsigned)n_cols);
                const unsigned q = ok ? q0 : q_safe;
                const unsigned px = (q * 10923u) >> 16;
                const unsigned row = px >= (unsigned)OW ? 1u : 0u, col = px - row * OW;
                oa[k] = (row * (unsigned)OWT + col) * 6u + (q - px * 6u);
                ov[k] = lds_read16(oimg + q * 16u);
                if constexpr (EMIT) xv3[k] = lds_read16(ximg + q * 16u);
            }
            lds_wait();
#pragma unroll
            for (int k = 0; k < 3; ++k) yo[oa[k]] = ov[k];
            if constexpr (EMIT) {
                uint4 *xo = reinterpret_cast<uint4 *>(p.t_out + (((long long)im * p.OH + oh0) * OWT + ow0) * COUT);
#pragma unroll
                for (int k = 0; k < 3; ++k) xo[oa[k]] = xv3[k];
            }
        }
        wg_barrier();   // also: the images are free for the next unit's rounds
        STAMP(9);
        ++g_units;
        unit = next_unit;
        {
            const uint32_t ns = lds_read4(next_slot_addr);
            lds_wait();
            next_unit = __builtin_amdgcn_readfirstlane((int)ns);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the patch prefetched for a unit that does not exist
}

constexpr int kRing2 = 256;
sc2_counter_ring g_ring2;
std::atomic<unsigned> g_seq2{0};

}  // namespace

extern "C" int sc2_conv2_gdn48_supported(int Cin, int Cout, int W) {
    return Cin == CIN && Cout == COUT && W >= 1 ? 1 : 0;   // (112: the static geometry; any other width: 56-column segments)
}

extern "C" int sc2_conv2_gdn48_fwd(const void *x, const void *w_frag, const void *gamma_frag, const float *beta, void *y, void *t_out,
                                   int N, int H, int W, int inverse, void *stream) {
    SC2_REQUIRE(x && w_frag && gamma_frag && beta && y, SC2_ERR_INVALID_ARG, "conv2_gdn48: null argument");
    SC2_REQUIRE(N > 0 && H > 0, SC2_ERR_INVALID_ARG, "conv2_gdn48: non-positive dimension");
    SC2_REQUIRE(sc2_conv2_gdn48_supported(CIN, COUT, W), SC2_ERR_UNSUPPORTED,
                "conv2_gdn48: needs a positive input width (got %d)", W);
    SC2_REQUIRE(((long long)(H + 2) * W + 2) * CIN * 2 < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv2_gdn48: image too large");
    Enc2Args a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.g = static_cast<const uint16_t *>(gamma_frag);
    a.beta = beta;
    a.y = static_cast<uint16_t *>(y);
    a.t_out = static_cast<uint16_t *>(t_out);
    a.N = N; a.H = H;
    a.OH = (H + 4 - 5) / 2 + 1;
    a.W = W; a.OWT = (W + 4 - 5) / 2 + 1;
    const bool seg = W != W_IN;
    a.n_seg = seg ? (a.OWT + OW - 1) / OW : 1;
    a.units_per_img = (a.OH + 1) / 2 * a.n_seg;
    const long long units = (long long)N * a.units_per_img;
    SC2_REQUIRE(units < 0x7FFFFFFFLL - 1024, SC2_ERR_UNSUPPORTED, "conv2_gdn48: too many units");
    a.n_units = (int)units;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<false, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<true, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2_gdn48_kernel<false, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int g_cus2 = sc2_device_cus();
    unsigned *slot = g_ring2.launch_slot(s, kRing2, 8, g_seq2);   // eight counters per launch
    if (!slot) return SC2_ERR_INTERNAL;
    const int grid = a.n_units < g_cus2 ? a.n_units : g_cus2;   // one 4-wave workgroup per CU
    a.unit_ctr = slot;
    a.stamps = nullptr;
#if SC2_ENC2_STAMPS
    const char *stamp_path = getenv("SC2_ENC2_STAMPS");
    const size_t stamp_bytes = 8 * 4 * 16 * N_STAMP * sizeof(unsigned long long);
    if (stamp_path) {
        void *sp = nullptr;
        (void)hipMalloc(&sp, stamp_bytes);
        (void)hipMemset(sp, 0, stamp_bytes);
        a.stamps = static_cast<unsigned long long *>(sp);
    }
#endif
#define SC2_ENC2_GO(INV, SEGM, EM) hipLaunchKernelGGL((conv2_gdn48_kernel<INV, SEGM, EM>), dim3(grid), dim3(256), LDS_BYTES, s, a)
    if (t_out) {   // training: y and the conv output in front of the GDN
        if (seg) { if (inverse) SC2_ENC2_GO(true, true, true); else SC2_ENC2_GO(false, true, true); }
        else { if (inverse) SC2_ENC2_GO(true, false, true); else SC2_ENC2_GO(false, false, true); }
    } else {
        if (seg) { if (inverse) SC2_ENC2_GO(true, true, false); else SC2_ENC2_GO(false, true, false); }
        else { if (inverse) SC2_ENC2_GO(true, false, false); else SC2_ENC2_GO(false, false, false); }
    }
#undef SC2_ENC2_GO
#if SC2_ENC2_STAMPS
    if (a.stamps) {
        (void)hipStreamSynchronize(s);
        unsigned long long *host = static_cast<unsigned long long *>(malloc(stamp_bytes));
        (void)hipMemcpy(host, a.stamps, stamp_bytes, hipMemcpyDeviceToHost);
        (void)hipFree(a.stamps);
        FILE *f = fopen(stamp_path, "wb");
        if (f) { fwrite(host, 1, stamp_bytes, f); fclose(f); }
        free(host);
    }
#endif
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
