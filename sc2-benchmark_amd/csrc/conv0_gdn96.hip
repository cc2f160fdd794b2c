// First encoder stage of the FP / SHP / MSHP bottlenecks in ONE persistent launch (gfx950):
//     y = GDN1_96( Conv2d(3 -> 96, k5, s2, p2, bias=False)(x) )               (sc2bench/models/layer.py:476-478)
// on the pixel-pair view of the image (bf16 [N, H, W/2, 8]: 2 pixels x 4 channels per 16 bytes; the 5-tap stride-2
// row filter is a 3-tap stride-1 filter over pairs, K = 5 x 3 x 8 = 120).
//
// The launch is a stream: 0.4 MB in, 2.4 MB out per image against 0.4 GFLOP.  On the generic tile kernel a workgroup
// does four k-slabs, the fused GDN and a 24 KB store, then waits for the acks: 2.5 TB/s.  Here 256-thread workgroups
// (two per CU, so one's epilogue overlaps the other's MFMAs) loop over UNITS of two output rows of one image
// (2 x 112 pixels x 96 channels = 43 KB out):
//   input   the 7 image rows a unit needs (12.5 KB) are staged in LDS, prefetched one unit ahead into registers
//           before the current unit's output stores are issued (vmcnt retires in issue order);
//   conv    A fragments straight from the staged rows (consecutive pixels = consecutive 16-byte pairs), W0 fragments
//           FRAGMENT-MAJOR from L2 into registers; wave (wm, wn) owns output row wm x channels [48 wn, 48 wn + 48);
//   GDN1    |t| (bf16) goes to an LDS image [224 px][96 ch] (rows padded to 208 B: conflict-free fragment reads),
//           norm = beta + gamma |t| is a second GEMM with gamma fragment-major from L2, y = t / norm on the f32
//           accumulators, written to the image and streamed out as one contiguous 43 KB block.
// Units are claimed with one atomic each, one unit ahead (claim c means unit c + 2 * gridDim.x; a launch makes exactly
// n_units claims, and the one that draws n_units - 1 writes the counter back to zero: no preset launch).  Geometry: any
// width -- an output row is cut into segments of OW = 112 pixels (one segment for the 224-pixel-wide input; 513 x 513 ->
// 257 = 112 + 112 + 33, 1216 -> 608 = 5 x 112 + 48), a unit is two output rows of one segment; an odd image width is
// zero-padded to even by the caller (the zero column is what the convolution's own padding would read).
// Round 5, PLANAR: the input is the reference's own f32 NCHW image batch, read where it lies -- a staged 16-byte chunk (pixel
// pair x 4 channels) is three 8-byte loads, one per colour plane, rounded to bf16 (round-to-nearest-even, as
// sc2_nchw_f32_to_nhwc_bf16 rounds) when the chunk is written to LDS: bit-identical to the layout pass + this kernel, without
// the pass (0.09 ms and 257 MB per 256-image batch) and without the [N, H, W, 4] bf16 copy.  Even widths only.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "sc2_common.h"

#ifndef SC2_ENC0_ROWS_EARLY
#define SC2_ENC0_ROWS_EARLY 1
#endif
#ifndef SC2_ENC0_PLANAR_EARLY
#define SC2_ENC0_PLANAR_EARLY 2   // PLANAR: how many of a thread's IN_Q chunks are fetched in front of the GDN phase
#endif
#ifndef SC2_NT_ENC0
#define SC2_NT_ENC0 1   // non-temporal output stores (sc2_common.h); 0: A/B
#endif

namespace {

struct EncArgs {
    const uint16_t *__restrict__ x;      // bf16 [N, H, WP, 8] pixel pairs; PLANAR: f32 [N, 3, H, 2 WP] (the NCHW image batch itself)
    const uint16_t *__restrict__ w;      // bf16 fragment-major [6][4][64][8]  (rows = 96 channels, K = 120 -> 128)
    const uint16_t *__restrict__ g;      // bf16 fragment-major gamma [6][3][64][8]
    const float *__restrict__ beta;      // f32 [96]
    uint16_t *__restrict__ y;            // bf16 NHWC [N, OH, OW, 96]
    uint16_t *__restrict__ t_out;        // EMIT instantiations: the conv output t in front of the GDN (bf16, laid out like y)
    int H, WP, OH, n_units, units_per_img;
    int n_seg;                           // column segments of OW output pixels per output row (WP = total output width)
    unsigned *unit_ctr;
};

constexpr int OW = 112, CH = 96, MT = OW / 16, NT = 3, KS1 = 4, KS2 = 3;
constexpr int IN_ROWS = 7, IN_PITCH = (OW + 2) * 16;               // staged input rows: pair columns -1 .. OW
constexpr int IN_BYTES = IN_ROWS * IN_PITCH;                        // 12 768
constexpr int IMG_PITCH = 208, IMG_BYTES = 2 * OW * IMG_PITCH;      // 46 592
constexpr int GAM_BYTES = 6 * KS2 * 64 * 16;                        // 18 432
constexpr int Y_Q = (2 * OW * (CH / 8) + 255) / 256;                     // output 16-byte chunks per thread (11)
constexpr int IN_Q = (IN_ROWS * (OW + 2) + 255) / 256;              // staged 16-byte chunks per thread (4)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(f32x2_t v) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// SEG = false: one 112-pixel segment per row (the 224-pixel-wide geometry), the unit's output is one contiguous block
// EMIT (round 5, the training forward): the conv output t leaves too -- the tensor the GDN's backward needs.  The LDS image holds
// |t| (the sign lives in the accumulators) and a second image does not fit beside two workgroups per CU, so t goes out from the
// registers, 8 bytes (a lane's four channels of a pixel) per store, when the image is written: the four quarter-lanes of a pixel
// fill one 32-byte sector per instruction and the three channel tiles of the wave the rest of its 96 bytes -- the L2 merges them.
template <bool INVERSE, bool SEG, bool PLANAR, bool EMIT = false>
__global__ __launch_bounds__(256, 2) void conv0_gdn96_kernel(const EncArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *rows = smem;
    unsigned char *img = smem + IN_BYTES;
    unsigned char *gam = smem + IN_BYTES + IMG_BYTES;   // gamma fragments [6][3][64] x 16 B, loaded once
    float *beta_s = reinterpret_cast<float *>(smem + IN_BYTES + IMG_BYTES + GAM_BYTES);   // [96]
    __shared__ int next_slot;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // output row of the unit, channel half
    const int frow = lane & 15, fq = lane >> 4;

    // Global loads of the tail are written branch-free (clamped address + select) so that the compiler counts vmcnt
    // exactly: its wait for the rows is then vmcnt(<stores issued after them>) and never waits for a store.
    uint4 in_next[PLANAR ? 1 : IN_Q];
    float2 in_pl[PLANAR ? IN_Q : 1][3];   // PLANAR: the chunk's pixel pair out of the three colour planes
    bool in_ok[IN_Q];
    auto load_rows = [&](int unit, int tid, auto k_lo, auto k_hi) {   // loads [k_lo, k_hi) of the 7 input rows of `unit`, zero outside the image
        const bool live = unit < p.n_units;
        const int im = live ? unit / p.units_per_img : 0;
        const int u_in = live ? unit - im * p.units_per_img : 0;
        const int rp = SEG ? u_in / p.n_seg : u_in, seg = SEG ? u_in - rp * p.n_seg : 0;
        const int oh0 = rp * 2;
        const uint16_t *ximg = p.x + (long long)im * p.H * p.WP * 8;
        [[maybe_unused]] const float *fimg = reinterpret_cast<const float *>(p.x) + (long long)im * 3 * p.H * p.WP * 2;
        [[maybe_unused]] const unsigned plane = (unsigned)(p.H * p.WP) * 8u;       // bytes of one colour plane
#pragma unroll
        for (int k = decltype(k_lo)::value; k < decltype(k_hi)::value; ++k) {
            const unsigned q = tid + 256 * k;
            const unsigned r = (q * 575u) >> 16, c = q - r * (OW + 2);      // q / 114 for q < 1100
            const int ih = 2 * oh0 - 2 + (int)r, pc = seg * OW + (int)c - 1;
            in_ok[k] = live & (q < IN_ROWS * (OW + 2)) & ((unsigned)ih < (unsigned)p.H) & ((unsigned)pc < (unsigned)p.WP);
            if constexpr (PLANAR) {
                const unsigned off = in_ok[k] ? (unsigned)(ih * p.WP + pc) * 8u : 0u;   // bytes within a plane (3 planes < 2^31)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    in_pl[k][ch] = *reinterpret_cast<const float2 *>(reinterpret_cast<const unsigned char *>(fimg) + ch * plane + off);
            } else {
                const unsigned off = in_ok[k] ? (unsigned)(ih * p.WP + pc) * 16u : 0u;   // bytes within the image (< 2^31)
                in_next[k] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(ximg) + off);
            }
        }
    };
    auto store_rows = [&](int tid) {
        // (PLANAR: the staged 16-byte chunk of a load = (pixel 0: c0 c1 c2 0)(pixel 1: c0 c1 c2 0), rounded to bf16 here)
        uint4 ck[IN_Q];
#pragma unroll
        for (int k = 0; k < IN_Q; ++k) {
            if constexpr (PLANAR) {
                ck[k].x = pack2(f32x2_t{in_pl[k][0].x, in_pl[k][1].x});
                ck[k].y = pack2(f32x2_t{in_pl[k][2].x, 0.f});
                ck[k].z = pack2(f32x2_t{in_pl[k][0].y, in_pl[k][1].y});
                ck[k].w = pack2(f32x2_t{in_pl[k][2].y, 0.f});
            } else {
                ck[k] = in_next[k];
            }
        }
#pragma unroll
        for (int k = 0; k < IN_Q; ++k) {
            const int q = tid + 256 * k;
            const bool inside = q < IN_ROWS * (OW + 2);       // k == IN_Q - 1: the others rewrite their previous chunk
            const int kk = k > 0 ? k - 1 : 0;
            const uint4 v = inside ? ck[k] : ck[kk];
            const bool ok = inside ? in_ok[k] : in_ok[kk];
            *reinterpret_cast<uint4 *>(rows + (inside ? q : q - 256) * 16) = ok ? v : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    const uint4 *wfrag = reinterpret_cast<const uint4 *>(p.w) + (long long)(wn * NT) * KS1 * 64;   // [(j*KS1 + ks)*64 + lane]

    // W0 fragments stay in registers for the life of the workgroup (48 VGPRs), gamma's and beta in LDS: no load is
    // issued behind the output stores
    uint4 wv[KS1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) wv[ks][j] = wfrag[(j * KS1 + ks) * 64 + lane];
    if (tid < CH) beta_s[tid] = p.beta[tid];
    for (int q = tid; q < GAM_BYTES / 16; q += 256)
        reinterpret_cast<uint4 *>(gam)[q] = reinterpret_cast<const uint4 *>(p.g)[q];
    const unsigned char *gfrag = gam + (wn * NT) * KS2 * 1024 + lane * 16;   // + (j*KS2 + ks) * 1024

    int unit = blockIdx.x;
    int next_unit = unit + gridDim.x;
    using K0 = std::integral_constant<int, 0>;
    using KN = std::integral_constant<int, IN_Q>;
    load_rows(unit, tid, K0{}, KN{});
    store_rows(tid);
    __syncthreads();

    while (unit < p.n_units) {
        const int im = unit / p.units_per_img;
        const int u_in = unit - im * p.units_per_img;
        const int rp = SEG ? u_in / p.n_seg : u_in, seg = SEG ? u_in - rp * p.n_seg : 0;
        const int oh0 = rp * 2;
        const int n_rows = p.OH - oh0 >= 2 ? 2 : 1;
        const int n_cols = p.WP - seg * OW >= OW ? OW : p.WP - seg * OW;   // valid output columns of this segment

        // ---------------------------------------------------------------- conv: t = W0 * patch
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int c = ks * 4 + fq;                 // 16-byte k chunk = (kh, pair tap t); chunk 15 is K padding
            const int kh = c / 3, t = c - kh * 3;
            const bool pad = c >= 15;
            const int a_lane = ((2 * wm + (pad ? 0 : kh)) * (OW + 2) + (pad ? 0 : t) + frow) * 16;   // + i * 256
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                // (the padding chunk reads tap (0, 0): finite image data against W0's zero K-padding columns)
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(rows + a_lane + i * 256));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[ks][j]), af,
                                                                        acc[i][j], 0, 0, 0);
            }
        }
        // |t| (bf16) of this lane's 4 channels of each accumulator tile -> image
        const int px_lane = wm * OW + frow;            // + i * 16
        [[maybe_unused]] unsigned char *t_px = nullptr;   // EMIT: this lane's pixel (row wm, column frow) of the unit in t_out, + i * 16 * 192
        if constexpr (EMIT) {
            int fr = frow, fk = fq;
            asm volatile("" : "+v"(fr), "+v"(fk));     // (per unit: nothing of this held across the units)
            const long long px0 = SEG ? ((long long)im * p.OH + oh0 + wm) * p.WP + seg * OW : ((long long)im * p.OH + oh0 + wm) * OW;
            t_px = reinterpret_cast<unsigned char *>(p.t_out) + (px0 + fr) * (CH * 2) + (wn * 48 + fk * 4) * 2;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wn * 48 + j * 16 + fq * 4;
                uint2 h;
                h.x = pack2(f32x2_t{acc[i][j][0], acc[i][j][1]});
                h.y = pack2(f32x2_t{acc[i][j][2], acc[i][j][3]});
                if constexpr (EMIT) {
                    if ((wm < n_rows) & (frow + i * 16 < n_cols)) *reinterpret_cast<uint2 *>(t_px + i * 16 * (CH * 2) + j * 32) = h;
                }
                h.x &= 0x7FFF7FFFu;
                h.y &= 0x7FFF7FFFu;
                *reinterpret_cast<uint2 *>(img + (px_lane + i * 16) * IMG_PITCH + col * 2) = h;
            }
        __syncthreads();   // staged rows consumed; |t| image complete (a pixel's 96 channels come from two waves)
        // the NEXT unit's seven rows are fetched here, in front of the GDN phase, instead of directly in front of this unit's
        // output stores (round 4: loads queued in front of a store burst delay it; conv2x2_gdn512 gained 5 % from the same move).
        // (The segmented-row instantiations keep the late fetch: with it here the forward one spilled.)
        // (PLANAR: a chunk is three 8-byte loads = 24 registers in flight instead of 16, which spilled: half of them go early,
        //  the other half stay directly in front of the output stores)
        constexpr int K_EARLY = (SC2_ENC0_ROWS_EARLY && !SEG) ? (PLANAR ? SC2_ENC0_PLANAR_EARLY : IN_Q) : 0;
        using KE = std::integral_constant<int, K_EARLY>;
        load_rows(next_unit, tid, K0{}, KE{});

        // ---------------------------------------------------------------- GDN1: norm = beta + gamma |t|
        // two passes over the pixel tiles (4 + 3): half the norm accumulators live at a time, and a pass's results may
        // overwrite its |t| rows as soon as both channel-half waves of the row have read them
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int MH = 4;
            const int i0 = half * MH;
            f32x4_t nrm[MH][NT];
#pragma unroll
            for (int i = 0; i < MH; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) nrm[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                uint4 gv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) gv[j] = *reinterpret_cast<const uint4 *>(gfrag + (j * KS2 + ks) * 1024);
#pragma unroll
                for (int i = 0; i < MH; ++i) {
                    if (i0 + i >= MT) continue;
                    const bf16x8_t xf = __builtin_bit_cast(
                        bf16x8_t,
                        *reinterpret_cast<const uint4 *>(img + (px_lane + (i0 + i) * 16) * IMG_PITCH + (ks * 4 + fq) * 16));
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        nrm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, gv[j]), xf,
                                                                            nrm[i][j], 0, 0, 0);
                }
            }
            __syncthreads();   // every wave has read this pass's |t| fragments: those rows may now take the results
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wn * 48 + j * 16 + fq * 4;
                const float4 b4 = *reinterpret_cast<const float4 *>(beta_s + col);
                const f32x2_t b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < MH; ++i) {
                    if (i0 + i >= MT) continue;
                    // explicit (e0,e1) / (e2,e3) pairs: packed add / mul / convert, no lane shuffles
                    const f32x2_t n01 = b01 + f32x2_t{nrm[i][j][0], nrm[i][j][1]};
                    const f32x2_t n23 = b23 + f32x2_t{nrm[i][j][2], nrm[i][j][3]};
                    const f32x2_t t01 = {acc[i0 + i][j][0], acc[i0 + i][j][1]}, t23 = {acc[i0 + i][j][2], acc[i0 + i][j][3]};
                    uint2 o;
                    if (INVERSE) {
                        o.x = pack2(t01 * n01);
                        o.y = pack2(t23 * n23);
                    } else {
                        o.x = pack2(t01 * f32x2_t{__builtin_amdgcn_rcpf(n01[0]), __builtin_amdgcn_rcpf(n01[1])});
                        o.y = pack2(t23 * f32x2_t{__builtin_amdgcn_rcpf(n23[0]), __builtin_amdgcn_rcpf(n23[1])});
                    }
                    *reinterpret_cast<uint2 *>(img + (px_lane + (i0 + i) * 16) * IMG_PITCH + col * 2) = o;
                }
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- next unit's rows, then stream this unit out
        // (the claim and the row loads are issued BEFORE the output stores: vmcnt retires in issue order, so waiting
        //  for them does not wait for the store acknowledgements)
        int tq = tid;   // opaque: the per-thread offsets below are recomputed, not kept (spilled) across the unit
        asm volatile("" : "+v"(tq));
        unsigned claimed = 0;
        if (tid == 0) {   // raw instruction: the compiler's atomicAdd waits for the result (vmcnt(0)) on the spot
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(p.unit_ctr), "v"(one) : "memory");
        }
        load_rows(next_unit, tq, KE{}, KN{});
        {
            if constexpr (!SEG) {
                uint4 *yo = reinterpret_cast<uint4 *>(p.y + ((long long)(im * p.OH + oh0) * OW) * CH);   // contiguous 2 rows
                const int n_chunks = n_rows * OW * (CH / 8);
#pragma unroll
                for (int k = 0; k < Y_Q; ++k) {
                    const unsigned q0 = tq + 256 * k;
                    const unsigned q = q0 < (unsigned)n_chunks ? q0 : (unsigned)tq;   // past the end: the thread's first chunk again
                    const unsigned px = (q * 43691u) >> 19;                           // q / 12 for q < 4096
                    if (SC2_NT_ENC0) sc2_store16_nt(yo + q, *reinterpret_cast<const uint4 *>(img + q * 16 + px * (IMG_PITCH - CH * 2)));
                    else yo[q] = *reinterpret_cast<const uint4 *>(img + q * 16 + px * (IMG_PITCH - CH * 2));
                }
            } else {
                // the unit's two output rows x n_cols pixels: runs of n_cols * 192 bytes at (oh0 + row, seg * OW); for the
                // 224-pixel-wide geometry (one segment, n_cols = OW) the two runs are one contiguous 43 KB block
                uint4 *yo = reinterpret_cast<uint4 *>(p.y + (((long long)im * p.OH + oh0) * p.WP + seg * OW) * CH);
                const unsigned q_safe = (unsigned)tq % (unsigned)(12 * n_cols);   // a chunk of row 0 that is always valid
#pragma unroll
                for (int k = 0; k < Y_Q; ++k) {
                    const unsigned q0 = tq + 256 * k;
                    const unsigned px0 = (q0 * 43691u) >> 19;                          // q0 / 12 for q0 < 4096
                    const unsigned row0 = px0 >= (unsigned)OW ? 1u : 0u, col0 = px0 - row0 * OW;
                    const bool ok = (row0 < (unsigned)n_rows) & (col0 < (unsigned)n_cols) & (q0 < (unsigned)(2 * OW * 12));
                    const unsigned q = ok ? q0 : q_safe;                               // invalid: a valid chunk again (same data)
                    const unsigned px = (q * 43691u) >> 19;
                    const unsigned row = px >= (unsigned)OW ? 1u : 0u, col = px - row * OW;
                    if (SC2_NT_ENC0) sc2_store16_nt(yo + (row * p.WP + col) * 12 + (q - px * 12), *reinterpret_cast<const uint4 *>(img + q * 16 + px * (IMG_PITCH - CH * 2)));
                    else yo[(row * p.WP + col) * 12 + (q - px * 12)] = *reinterpret_cast<const uint4 *>(img + q * 16 + px * (IMG_PITCH - CH * 2));
                }
            }
        }
        store_rows(tq);    // the staged rows were last read before the first barrier of this unit
        if (tid == 0) {    // the claim is older than the row loads store_rows() has just waited for
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(claimed) : "n"(Y_Q) : "memory");
            next_slot = (int)(claimed + 2 * gridDim.x);
            if (claimed == (unsigned)(p.n_units - 1)) *p.unit_ctr = 0u;   // the launch's last claim re-arms the counter
        }
        __syncthreads();   // next rows visible; image free
        unit = next_unit;
        next_unit = __builtin_amdgcn_readfirstlane(next_slot);
    }
}

constexpr int kRing0 = 256;
sc2_counter_ring g_ring0;
std::atomic<unsigned> g_seq0{0};

}  // namespace

extern "C" int sc2_conv0_gdn96_supported(int Cin_pairs, int Cout, int W_pairs) {
    return Cin_pairs == 8 && Cout == 96 && W_pairs >= 1 ? 1 : 0;
}

namespace {
template <bool PLANAR, bool EMIT>
int launch_conv0_gdn96(const void *x, const void *w_frag, const void *gamma_frag, const float *beta, void *y, void *t_out, int N, int H,
                       int W_pairs, int inverse, void *stream) {
    SC2_REQUIRE(x && w_frag && gamma_frag && beta && y, SC2_ERR_INVALID_ARG, "conv0_gdn96: null argument");
    SC2_REQUIRE(N > 0 && H > 0, SC2_ERR_INVALID_ARG, "conv0_gdn96: non-positive dimension");
    SC2_REQUIRE(sc2_conv0_gdn96_supported(8, 96, W_pairs), SC2_ERR_UNSUPPORTED,
                "conv0_gdn96: needs at least one pixel pair per row, got %d", W_pairs);
    // (per-image byte offsets are 32-bit: the pair view is 16 B per pair, the f32 planes 3 x 8 B per pair)
    SC2_REQUIRE((long long)H * W_pairs * (PLANAR ? 24 : 16) < 0x7FFFFFFFLL, SC2_ERR_UNSUPPORTED, "conv0_gdn96: image too large");
    EncArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.g = static_cast<const uint16_t *>(gamma_frag);
    a.beta = beta;
    a.y = static_cast<uint16_t *>(y);
    a.t_out = static_cast<uint16_t *>(t_out);
    a.H = H; a.WP = W_pairs;
    a.OH = (H + 4 - 5) / 2 + 1;
    a.n_seg = (W_pairs + OW - 1) / OW;
    a.units_per_img = ((a.OH + 1) / 2) * a.n_seg;
    const long long units = (long long)N * a.units_per_img;
    SC2_REQUIRE(units < 0x7FFFFFFFLL - 1024, SC2_ERR_UNSUPPORTED, "conv0_gdn96: too many units");
    a.n_units = (int)units;
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int lds = IN_BYTES + IMG_BYTES + GAM_BYTES + CH * 4;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn96_kernel<true, false, PLANAR, EMIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn96_kernel<false, false, PLANAR, EMIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn96_kernel<true, true, PLANAR, EMIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn96_kernel<false, true, PLANAR, EMIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int cus0 = sc2_device_cus();
    unsigned *slot = g_ring0.launch_slot(static_cast<hipStream_t>(stream), kRing0, 1, g_seq0);
    if (!slot) return SC2_ERR_INTERNAL;
    const int grid = a.n_units < 2 * cus0 ? a.n_units : 2 * cus0;   // two workgroups per CU
    a.unit_ctr = slot;
    const bool seg = W_pairs != OW;
    if (inverse && seg) hipLaunchKernelGGL((conv0_gdn96_kernel<true, true, PLANAR, EMIT>), dim3(grid), dim3(256), lds, s, a);
    else if (inverse) hipLaunchKernelGGL((conv0_gdn96_kernel<true, false, PLANAR, EMIT>), dim3(grid), dim3(256), lds, s, a);
    else if (seg) hipLaunchKernelGGL((conv0_gdn96_kernel<false, true, PLANAR, EMIT>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv0_gdn96_kernel<false, false, PLANAR, EMIT>), dim3(grid), dim3(256), lds, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
}  // namespace

extern "C" int sc2_conv0_gdn96_fwd(const void *x_pairs, const void *w_frag, const void *gamma_frag, const float *beta,
                                   void *y, void *t_out, int N, int H, int W_pairs, int inverse, void *stream) {
    if (t_out) return launch_conv0_gdn96<false, true>(x_pairs, w_frag, gamma_frag, beta, y, t_out, N, H, W_pairs, inverse, stream);
    return launch_conv0_gdn96<false, false>(x_pairs, w_frag, gamma_frag, beta, y, nullptr, N, H, W_pairs, inverse, stream);
}

extern "C" int sc2_conv0_gdn96_nchw_fwd(const float *x_nchw, const void *w_frag, const void *gamma_frag, const float *beta,
                                        void *y, int N, int H, int W, int inverse, void *stream) {
    SC2_REQUIRE(W > 0 && W % 2 == 0, SC2_ERR_UNSUPPORTED,
                "conv0_gdn96_nchw: even widths only (got %d): a row of an odd-width plane does not start on a pixel pair", W);
    return launch_conv0_gdn96<true, false>(x_nchw, w_frag, gamma_frag, beta, y, nullptr, N, H, W / 2, inverse, stream);
}
