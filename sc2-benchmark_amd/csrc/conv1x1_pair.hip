// Two consecutive 1x1 layers of the ResNet tail across a block boundary, in ONE launch (gfx950):
//     h = relu(W3 o + b3 + identity)        conv3 + bn3 + residual + ReLU of torchvision Bottleneck block b   (C  <- K1)
//     u = relu(W1 h + b1)                   conv1 + bn1 + ReLU of block b + 1                                 (N2 <- C)
// (sc2bench/models/backbone.py:235-254 runs them as consecutive blocks of layer2; BatchNorm folded, eval mode.)
//
// Why: both are pure HBM streams on their own -- conv3 reads o (51 MB at bs 256) + the identity (205 MB) and writes h (205 MB) at
// 4.4 TB/s, conv1 of the next block reads h AGAIN (205 MB) to write 51 MB: 0.103 + 0.067 ms per block of layer2, four and three
// times per step.  Here a tile holds ALL C channels of its pixels, so h goes to HBM once (it is the next block's identity) and
// feeds the second GEMM from LDS: 718 MB -> 513 MB per block pair.
//
// Structure: persistent 512-thread workgroups (one per CU: 140 KB of LDS), tiles of P = 112 consecutive pixels (7 MFMA row tiles;
// 200 704 pixels = 1 792 tiles = 7 per CU), claimed with one atomic per tile, one tile ahead.
//   identity  the tile's [P][C] bf16 rows go global -> LDS IMAGE directly (buffer-addressed LDS-DMA, one 1 KB row per
//             wave-instruction; the lane -> chunk XOR swizzle is applied on the SOURCE side), issued as soon as the image is free
//             and waited for after phase 1
//   phase 1   acc = W3 o: o tile [P][K1] in LDS (KS1 slabs of [P][64 B]), W3 fragments resident in registers for the life of
//             the workgroup (wave w owns channels [C/8 w, C/8 (w + 1)) of every pixel; weights as the MFMA A operand, so a lane
//             holds 4 consecutive channels of a pixel = one 8-byte image slot)
//   epilogue1 h = relu((acc + b3) + identity) in place in the image (the operation order of the streaming kernel)
//   phase 2   acc2 = W1 h: K = C from the image (every wave reads all rows), wave w owns 16 of the N2 channels, W1 fragments
//             stream from L2 (fragment-major, two steps ahead); the image is streamed out to HBM (h) meanwhile
//   epilogue2 u = relu(acc2 + b1) -> bf16 staging in the image region (free once h sits in registers for its stores) ->
//             coalesced 16-byte stores; N2 = 128 (conv1 of the next layer2 block) or 256 (conv1 of layer3.0 behind layer2.3)
#include <stdlib.h>

#include <atomic>

#include "sc2_common.h"

#ifndef SC2_NT_PAIR
#define SC2_NT_PAIR 0   // non-temporal output stores: measured SLOWER here (the consumer launch finds part of this map in L2 / the memory-side cache: head + 2.5 %, dec.conv2 + 2 %); 1: A/B
#endif

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// (the wait states behind a 16-byte buffer store: see buf_store16 in conv2x2_win.hip)
__device__ __forceinline__ void buf_store16(buf_rsrc_t r, uint32_t voff, uint32_t soff, u32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, SC2_NT_PAIR ? SC2_BUF_AUX_NT : 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#else   // host pass: stand-ins
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
__device__ __forceinline__ uint4 buf_load16(buf_rsrc_t, uint32_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void buf_store16(buf_rsrc_t, uint32_t, uint32_t, u32x4_t) {}
#endif

// LDS accesses the compiler must not see: hipcc knows that an LDS-DMA load writes LDS and drains vmcnt(0) in front of every LDS
// access it can see behind one (may-alias); between issuing the next tile's identity rows and the counted wait for them, the o
// tile is written and read through these
template <int OFF>
__device__ __forceinline__ u32x4_t lds_read16_imm(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write16(uint32_t addr, u32x4_t v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

#ifndef SC2_PAIR_DBG
#define SC2_PAIR_DBG 0   // timing experiments (results garbage): 1 = no identity loads, 2 = no output stores, 4 = no second GEMM
#endif

struct PairArgs {
    const uint16_t *__restrict__ o;      // bf16 [M][K1]   input of conv3 (block b's conv2 output)
    const uint16_t *__restrict__ w3;     // bf16 fragment-major [C/16][K1/32][64][8]
    const float *__restrict__ b3;        // f32 [C]
    const uint16_t *__restrict__ idn;    // bf16 [M][C]    identity of block b
    uint16_t *__restrict__ h;            // bf16 [M][C]    block b's output
    const uint16_t *__restrict__ w1;     // bf16 fragment-major [N2/16][C/32][64][8]
    const float *__restrict__ b1;        // f32 [N2]
    uint16_t *__restrict__ u;            // bf16 [M][N2]   conv1 output of block b + 1
    int M, n_tiles;
    int rev;                             // 1: tiles are walked from the END of the map (see sc2_conv1x1_pair_fwd)
    unsigned *tile_ctr;                  // claims so far (claim c = tile c + 2 * gridDim.x); zero between launches
};

template <int C, int K1, int N2, int MT>
__global__ __launch_bounds__(512, 2) void conv1x1_pair_kernel(const PairArgs p) {
    constexpr int P = MT * 16;                 // pixels per tile
    constexpr int KS1 = K1 / 32, KS2 = C / 32;
    constexpr int NT1 = C / 8 / 16;            // 16-channel tiles of a wave in phase 1
    constexpr int ROWB = C * 2;                // image row bytes
    constexpr int CPR = ROWB / 16;             // 16-byte chunks per image row
    constexpr int IMG = P * ROWB, SLAB = P * 64, OT = KS1 * SLAB;
    constexpr int NT2 = N2 / 128;              // 16-channel tiles of u per wave (wave w owns channels [N2/8 w, N2/8 (w + 1)))
    static_assert(N2 == 128 || N2 == 256, "one or two 16-channel tiles of u per wave");
    static_assert(N2 <= C, "the u staging lives in the image region");
    static_assert(CPR == 64, "one image row = one LDS-DMA wave-instruction");
    constexpr int UROW = N2 * 2, UCPR = UROW / 16;        // u staging row bytes / 16-byte chunks per row
    constexpr int OQ = (P * K1 * 2 / 16 + 511) / 512;     // 16-byte chunks of an o tile per thread
    constexpr int UQ = (P * UCPR + 511) / 512;            // ... of a u tile
    constexpr int IQ = (P * CPR + 511) / 512;             // image chunks per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *img = smem;
    unsigned char *ot = smem + IMG;
    volatile int *next_slot = reinterpret_cast<volatile int *>(smem + IMG + OT);
    float *bias_lds = reinterpret_cast<float *>(smem + IMG + OT + 16);      // [C] b3, then [N2] b1: read in the epilogues from LDS,
                                                                            // so that no epilogue waits on the vector-memory counter

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

    constexpr uint32_t OOB = 0x80000000u;
    const buf_rsrc_t rs_idn = make_rsrc(p.idn, (uint32_t)((long long)p.M * ROWB));
    const buf_rsrc_t rs_o = make_rsrc(p.o, (uint32_t)((long long)p.M * K1 * 2));
    const buf_rsrc_t rs_h = make_rsrc(p.h, (uint32_t)((long long)p.M * ROWB));
    const buf_rsrc_t rs_u = make_rsrc(p.u, (uint32_t)((long long)p.M * N2 * 2));

    // ---- helpers ---------------------------------------------------------------------------------------------
    // o tile: thread chunk q = tid + 512 k -> (row = q / (K1/8), c16 = q % (K1/8)) -> slab c16 / 4, chunk c16 % 4
    auto load_o = [&](int tile, uint4 (&ov)[OQ], int tq) {
        // straight-line: every load is issued, masked lanes / tiles go out of range and read zeros, so that the compiler's vmcnt
        // bookkeeping stays exact (a conditional load turned into a branch made it wait vmcnt(0) -- for the output stores)
        const bool t_ok = tile < p.n_tiles;
        const int tpos = p.rev ? p.n_tiles - 1 - tile : tile;
        const uint32_t so = (uint32_t)(t_ok ? (long long)tpos * P * (K1 * 2) : 0);
#pragma unroll
        for (int k = 0; k < OQ; ++k) {
            const int q = tq + 512 * k;
            const bool ok = t_ok && q < P * (K1 / 8) && (long long)tpos * P + q / (K1 / 8) < p.M;
            ov[k] = buf_load16(rs_o, ok ? (uint32_t)q * 16u : OOB, so);
        }
    };
    auto store_o = [&](const uint4 (&ov)[OQ], int tq) {
#pragma unroll
        for (int k = 0; k < OQ; ++k) {
            const int q = tq + 512 * k;
            if (q < P * (K1 / 8)) {
                const int row = q / (K1 / 8), c16 = q % (K1 / 8);
                lds_write16(lds_base + IMG + (c16 >> 2) * SLAB + row * 64 + (((c16 & 3) ^ ((row >> 1) & 3)) << 4),
                            __builtin_bit_cast(u32x4_t, ov[k]));
            }
        }
    };
    // identity rows of `tile` -> image: wave w takes rows w, w + 8, ...; LDS chunk position `lane` of a row holds global chunk
    // lane ^ (row & 15) (the image's swizzle, applied on the source side); rows past M read zeros (out of range)
    auto issue_identity = [&](int tile, int lq) {
        static_assert(P % 8 == 0, "every wave takes P / 8 rows: straight-line issue, exact vmcnt bookkeeping");
#pragma unroll
        for (int r = 0; r < P / 8; ++r) {
            const int row = wn + 8 * r;
            const long long m = (long long)(p.rev ? p.n_tiles - 1 - tile : tile) * P + row;
            const bool ok = tile < p.n_tiles && m < p.M && !(SC2_PAIR_DBG & 1);      // wave-uniform
            const uint32_t voff = ok ? (uint32_t)((lq ^ (row & 15)) << 4) : OOB;
            buf_load_lds16(rs_idn, (lds_ptr_t)(img + row * ROWB), voff, (uint32_t)(ok ? m * ROWB : 0));
        }
    };

    // ---- per-workgroup constants -------------------------------------------------------------------------------
    // W3 fragments of this wave (channel tiles wn * NT1 + j, all K1): 64 registers, resident for the life of the workgroup (phase 2
    // parks its sixteen W1 fragments in the accumulators' registers, which are dead by then)
    // (buffer-addressed: ONE per-lane offset register + a scalar offset per fragment; as 16 global pointers the addresses were
    //  hoisted out of the tile loop, spilled, and every reload waited vmcnt(0) -- for the identity rows just issued)
    uint4 wv[KS1][NT1];
    const buf_rsrc_t rs_w3 = make_rsrc(p.w3, (uint32_t)(C * K1 * 2));
    const uint32_t w3_vo = (uint32_t)(((wn * NT1) * KS1 * 64 + lane) * 16);
    auto load_w = [&]() {
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int s = 0; s < KS1; ++s) wv[s][j] = buf_load16(rs_w3, w3_vo, (uint32_t)((j * KS1 + s) * 1024));
    };
    load_w();
    // (per-lane LDS offsets are formed inside the tile loop from an opaque copy of the lane index: the tile-dependent part of
    //  every address is a compile-time constant -- row = 16 i + frow, so row & 15 = frow and (row >> 1) & 3 = (frow >> 1) & 3 --
    //  and what is loop-invariant would be hoisted out of the loop and held in ~50 registers across it)
    const buf_rsrc_t rs_w1 = make_rsrc(p.w1, (uint32_t)(N2 * C * 2));
    const uint32_t w1_vo = (uint32_t)(((wn * NT2) * KS2 * 64 + lane) * 16);      // + (t * KS2 + ks) * 1024

    for (int k = tid; k < C + N2; k += 512) bias_lds[k] = k < C ? p.b3[k] : p.b1[k - C];
    int tile = blockIdx.x;
    int next_tile = tile + gridDim.x;
    {
        uint4 ov[OQ];
        load_o(tile, ov, tid);
        store_o(ov, tid);
        issue_identity(tile, lane);
    }
    __syncthreads();

    bool first = true;
    while (tile < p.n_tiles) {
        const long long m0 = (long long)(p.rev ? p.n_tiles - 1 - tile : tile) * P;
        // opaque copies of the thread / lane index: the per-thread offsets of the copy loops below are recomputed inside the tile
        // loop instead of being hoisted out of it and spilled (hipcc kept ~70 of them in scratch and reloaded them between the
        // output stores)
        int tq = tid, lq = lane;
        asm volatile("" : "+v"(tq), "+v"(lq));
        const int frow = lq & 15, fq = lq >> 4;                                  // (shadow the hoistable ones)
        const int o_lane = frow * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);          // + s * SLAB + i * 1024
        const int img_row = frow * ROWB;                                           // + i * 16 * ROWB
        // the claim of the tile after next: a raw instruction (the compiler's atomicAdd waits for the result, vmcnt(0), on the
        // spot), issued here and read behind phase 2 -- 256 workgroups on one counter take 1 - 3 us to answer
        unsigned claimed = 0;
        if (tid == 0) {
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(p.tile_ctr), "v"(one) : "memory");
        }
        // ------------------------------------------------------------------ phase 1: acc = W3 o
        f32x4_t acc[MT][NT1];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            const uint32_t o_addr = lds_base + IMG + o_lane;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                // row tiles in two batches (4 + 3): seven fragments in flight at once cost 28 registers on top of the 176 of
                // accumulators + weights, and hipcc spilled five weight fragments
                u32x4_t xa[4], xb[3];
                static_assert(MT == 7, "the asm waits below name 4 + 3 registers");
                xa[0] = lds_read16_imm<0 * 1024>(o_addr + s * SLAB);
                xa[1] = lds_read16_imm<1 * 1024>(o_addr + s * SLAB);
                xa[2] = lds_read16_imm<2 * 1024>(o_addr + s * SLAB);
                xa[3] = lds_read16_imm<3 * 1024>(o_addr + s * SLAB);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3])::"memory");
                xb[0] = lds_read16_imm<4 * 1024>(o_addr + s * SLAB);
                xb[1] = lds_read16_imm<5 * 1024>(o_addr + s * SLAB);
                xb[2] = lds_read16_imm<6 * 1024>(o_addr + s * SLAB);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, xa[i]);
#pragma unroll
                    for (int j = 0; j < NT1; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[s][j]), xf, acc[i][j], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2])::"memory");
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, xb[i]);
#pragma unroll
                    for (int j = 0; j < NT1; ++j)
                        acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[s][j]), xf, acc[4 + i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // this wave's identity rows have landed.  They were issued BEFORE the previous tile's output stores (vmcnt retires in
        // issue order), so the wait leaves exactly those IQ + UQ stores outstanding: reads of this tile never wait for the write
        // stream of the last one.  (Every store below is issued unconditionally -- masked lanes go out of range -- so the count
        // is exact.)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wn == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IQ + UQ + 1) : "memory");   // (+ this tile's claim, wave 0 only)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IQ + UQ) : "memory");
        first = false;
        __syncthreads();                                     // ... everybody's; the o tile has been consumed
        // ------------------------------------------------------------------ epilogue 1: h = relu((acc + b3) + identity), in place
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            const int c16 = wn * (NT1 * 2) + j * 2 + (fq >> 1);
            const float4 b3v = *reinterpret_cast<const float4 *>(bias_lds + wn * (NT1 * 16) + j * 16 + fq * 4);
            const f32x2_t b01 = {b3v.x, b3v.y}, b23 = {b3v.z, b3v.w};
            const int slot_lane = img_row + ((c16 ^ frow) << 4) + (fq & 1) * 8;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                unsigned char *slot = img + slot_lane + i * (16 * ROWB);
                const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]} + b01;
                f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]} + b23;
                v01 += f32x2_t{__builtin_bit_cast(float, xr.x << 16), __builtin_bit_cast(float, xr.x & 0xFFFF0000u)};
                v23 += f32x2_t{__builtin_bit_cast(float, xr.y << 16), __builtin_bit_cast(float, xr.y & 0xFFFF0000u)};
                v01 = f32x2_t{fmaxf(v01[0], 0.f), fmaxf(v01[1], 0.f)};
                v23 = f32x2_t{fmaxf(v23[0], 0.f), fmaxf(v23[1], 0.f)};
                uint2 o2;
                o2.x = pack2(v01);
                o2.y = pack2(v23);
                *reinterpret_cast<uint2 *>(slot) = o2;
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // the image holds h for all C channels of the tile
        // ------------------------------------------------------------------ next tile's o rows, the claim after next
        // W1 fragments of this wave, EIGHT k-steps ahead, into registers the accumulators have just left: two steps of read-ahead
        // left every k-step of phase 2 (7 MFMAs) waiting ~600 cycles for its fragment to come from L2 (timing experiment: the
        // phase took 10 k cycles per tile against 1.8 k of MFMA issue)
        __builtin_amdgcn_sched_barrier(0);   // (not above the epilogue: the accumulators' registers are what they go into)
        constexpr int GB = 8;     // k-steps in flight: eight (8 x 7 MFMAs per channel tile ~ 900 cycles) cover the L2 round trip
        uint4 gb[GB][NT2];
#pragma unroll
        for (int ks = 0; ks < GB; ++ks)
#pragma unroll
            for (int t = 0; t < NT2; ++t) gb[ks][t] = buf_load16(rs_w1, w1_vo, (uint32_t)((t * KS2 + ks) * 1024));
        uint4 ov[OQ];
        load_o(next_tile, ov, tq);
        // ------------------------------------------------------------------ phase 2: acc2 = W1 h
        f32x4_t acc2[MT][NT2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT2; ++t) acc2[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ((SC2_PAIR_DBG & 4) ? 0 : KS2); ++ks) {
            bf16x8_t gf[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                gf[t] = __builtin_bit_cast(bf16x8_t, gb[ks % GB][t]);
                if (ks + GB < KS2) gb[ks % GB][t] = buf_load16(rs_w1, w1_vo, (uint32_t)((t * KS2 + ks + GB) * 1024));
            }
            const int rd_lane = img_row + (((ks * 4 + fq) ^ frow) << 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(img + rd_lane + i * (16 * ROWB)));
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], xf, acc2[i][t], 0, 0, 0);
            }
            if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (tid == 0) {    // the claim is older than everything issued since, bar the OQ loads of the next o tile
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(claimed) : "n"(OQ) : "memory");
            *next_slot = (int)(claimed + 2 * gridDim.x);
            if (claimed == (unsigned)(p.n_tiles - 1)) *p.tile_ctr = 0u;   // the launch's last claim re-arms the counter
        }
        // ------------------------------------------------------------------ h: the image into registers (stored further down)
        u32x4_t hv[IQ];
#pragma unroll
        for (int k = 0; k < IQ; ++k) {
            const int q = tq + 512 * k;
            const int row = q / CPR, c = q % CPR;
            hv[k] = *reinterpret_cast<const u32x4_t *>(img + row * ROWB + ((c ^ (row & 15)) << 4));
        }
        __syncthreads();   // every wave is done with the image (its region now stages u); the o region has been free since phase 1
        store_o(ov, tq);   // the next tile's o rows (asm ds_write: see lds_read16_imm)
        // ------------------------------------------------------------------ epilogue 2: u = relu(acc2 + b1) -> staging [P][N2]
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            const float4 b1v = *reinterpret_cast<const float4 *>(bias_lds + C + (wn * NT2 + t) * 16 + fq * 4);
            const f32x2_t b01 = {b1v.x, b1v.y}, b23 = {b1v.z, b1v.w};
            const int c16 = (wn * NT2 + t) * 2 + (fq >> 1);
            const int u_lane = frow * UROW + ((c16 ^ frow) << 4) + (fq & 1) * 8;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x2_t v01 = f32x2_t{acc2[i][t][0], acc2[i][t][1]} + b01;
                f32x2_t v23 = f32x2_t{acc2[i][t][2], acc2[i][t][3]} + b23;
                v01 = f32x2_t{fmaxf(v01[0], 0.f), fmaxf(v01[1], 0.f)};
                v23 = f32x2_t{fmaxf(v23[0], 0.f), fmaxf(v23[1], 0.f)};
                uint2 o2;
                o2.x = pack2(v01);
                o2.y = pack2(v23);
                *reinterpret_cast<uint2 *>(img + u_lane + i * (16 * UROW)) = o2;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the asm writes of store_o)
        __syncthreads();   // u staged; the next o tile is visible
        const int tile_after_next = __builtin_amdgcn_readfirstlane(*next_slot);
        u32x4_t uv[UQ];
#pragma unroll
        for (int k = 0; k < UQ; ++k) {
            const int q = tq + 512 * k;
            const int row = q / UCPR, c = q % UCPR;
            const int qq = q < P * UCPR ? row * UROW + ((c ^ (row & 15)) << 4) : 0;
            uv[k] = *reinterpret_cast<const u32x4_t *>(img + qq);
        }
        __syncthreads();   // the staging has been read by everybody: the image may take the next tile's identity rows
        // From here to the counted wait behind the next phase 1 no LDS access is visible to the compiler (see lds_read16_imm).
        issue_identity(next_tile, lq);     // ... which start to arrive now,
        // and only then this tile's output goes out: h from the registers (the tile is contiguous in HBM), then u
        {
            const uint32_t so = (uint32_t)(m0 * ROWB);
#pragma unroll
            for (int k = 0; k < IQ; ++k) {
                const int q = tq + 512 * k;
                buf_store16(rs_h, (q < P * CPR && m0 + q / CPR < p.M && !(SC2_PAIR_DBG & 2)) ? (uint32_t)q * 16u : OOB, so, hv[k]);
            }
        }
        {
            const uint32_t so = (uint32_t)(m0 * UROW);
#pragma unroll
            for (int k = 0; k < UQ; ++k) {
                const int q = tq + 512 * k;
                buf_store16(rs_u, (q < P * UCPR && m0 + q / UCPR < p.M && !(SC2_PAIR_DBG & 2)) ? (uint32_t)q * 16u : OOB, so, uv[k]);
            }
        }
        tile = next_tile;
        next_tile = tile_after_next;
    }
}

constexpr int kRing = 256;
sc2_counter_ring g_ring;
std::atomic<unsigned> g_seq{0};

template <int C, int K1, int N2, int MT>
int launch_pair(const PairArgs &a, hipStream_t s) {
    constexpr int P = MT * 16;
    constexpr int lds = P * C * 2 + (K1 / 32) * P * 64 + 16 + (C + N2) * 4;
    static_assert(lds <= 160 * 1024, "image + o tile must fit the CU's LDS");
    // per DEVICE, like the tile counters: the dynamic-LDS attribute of a function is a property of the device's code object, and
    // a process may launch on several devices
    static bool attr_set[SC2_MAX_DEVICES] = {};
    const int dev = sc2_device_slot();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1x1_pair_kernel<C, K1, N2, MT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    const int g_cus = sc2_device_cus();
    unsigned *slot = g_ring.launch_slot(s, kRing, 1, g_seq);
    if (!slot) return SC2_ERR_INTERNAL;
    PairArgs b = a;
    b.n_tiles = (a.M + P - 1) / P;
    b.tile_ctr = slot;
    {
        // Every other launch walks its tiles from the END of the map (SC2_PAIR_ALT=0: always front to back).  The block input a
        // launch re-reads as the identity was written by the previous pair launch front to back; read front to back again, its
        // head has been pushed out of the 256 MB memory-side cache by that very traffic (LRU, 512 MB per launch), read back to
        // front its newest part is still there.  Small: the four launches of a step 0.549 -> 0.536 ms, head - 0.5 % (the
        // 3x3 layer between two pair launches and the pair's own writes leave ~100 MB of the identity in the cache).
        const int alt = sc2_pol().pair_alt;
        static std::atomic<unsigned> calls{0};
        b.rev = alt ? (int)(calls.fetch_add(1) & 1u) : 0;
    }
    const int grid = b.n_tiles < g_cus ? b.n_tiles : g_cus;
    hipLaunchKernelGGL((conv1x1_pair_kernel<C, K1, N2, MT>), dim3(grid), dim3(512), lds, s, b);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv1x1_pair_supported(int K1, int C, int N2) { return (K1 == 128 && C == 512 && (N2 == 128 || N2 == 256)) ? 1 : 0; }

extern "C" int sc2_conv1x1_pair_fwd(const void *o, const void *w3_frag, const float *b3, const void *identity, void *h,
                                    const void *w1_frag, const float *b1, void *u, long long M, int K1, int C, int N2,
                                    void *stream) {
    SC2_REQUIRE(o && w3_frag && b3 && identity && h && w1_frag && b1 && u, SC2_ERR_INVALID_ARG, "conv1x1_pair: null argument");
    SC2_REQUIRE(sc2_conv1x1_pair_supported(K1, C, N2), SC2_ERR_UNSUPPORTED, "conv1x1_pair: K1 %d C %d N2 %d not supported", K1, C, N2);
    SC2_REQUIRE(M > 0 && M * C * 2 < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv1x1_pair: M = %lld outside the 32-bit buffer range", M);
    PairArgs a;
    a.o = static_cast<const uint16_t *>(o); a.w3 = static_cast<const uint16_t *>(w3_frag); a.b3 = b3;
    a.idn = static_cast<const uint16_t *>(identity); a.h = static_cast<uint16_t *>(h);
    a.w1 = static_cast<const uint16_t *>(w1_frag); a.b1 = b1; a.u = static_cast<uint16_t *>(u);
    a.M = (int)M; a.n_tiles = 0; a.tile_ctr = nullptr; a.rev = 0;
    if (N2 == 256) return launch_pair<512, 128, 256, 7>(a, static_cast<hipStream_t>(stream));
    return launch_pair<512, 128, 128, 7>(a, static_cast<hipStream_t>(stream));
}
