// Reference-precision convolution / GDN: f32 operands on the f32 matrix cores.
//
// The reference's CPU path runs the analysis transform in f32 (nn.Conv2d + CompressAI GDN1, sc2bench/models/layer.py:475-483);
// the symbols it then codes are round(latent - median) (layer.py:506), so the byte stream of an image depends on every bit of
// the latent that lies near a rounding boundary.  The bf16 MFMA kernels (conv_igemm*, conv0_gdn96, conv2_gdn48, conv2x2_c48)
// move ~1-2 % of the symbols of an image across such a boundary.  This file is the encoder mode that does not:
// `v_mfma_f32_16x16x4_f32` takes f32 A / B operands and accumulates in f32 -- bit for bit a k-ordered fmaf chain, one rounding
// per product (cdna_hip_programming.md, "FP32-input MFMA") -- at 1/16 of the bf16 matrix rate (155 TFLOP/s).  The encoder is
// only 1.18 GFLOP per image, so the mode costs ~3 ms per 256 images.  What remains against the CPU's f32 convolution is the
// ORDER of the f32 additions (the CPU's blocked loops sum in another order): relative 1e-7 per element, symbol flips ~1e-5.
//
// Structure:
//   * implicit GEMM, M = output pixels, N = output channels, K = (kh, kw, ci) with ci fastest; activations f32 NHWC with the
//     channel count padded to a multiple of 4, so a lane's four consecutive k are ONE 16-byte load of one input pixel;
//   * a k-step = 16 consecutive k = 4 MFMAs: lane (row r = l & 15, quarter q = l >> 4) loads k = 16 s + 4 q + j (j = 0..3) of
//     its pixel / its output channel and MFMA j consumes element j of every lane (k = 16 s + 4 q + j for q = 0..3): the same
//     permutation on both operands, so every product lands in the right sum;
//   * a wave owns MT x 16 pixels x NT x 16 channels; weights are packed fragment-major ([chunk][step][nt][lane][4], 1 KB per
//     fragment) by hip.pack_conv_f32;
//   * round 4: the four waves of a workgroup read the SAME weight stream, so it goes through LDS once per workgroup (direct-to-LDS
//     loads, a two-deep ring of four-step groups, one barrier per group); the activations stay per-wave register loads, two
//     steps ahead, issued and counted by hand (asm; tools/audit_vmcnt.py --copies checks that nothing touches a register whose
//     load is in flight).  Measured on the 96 -> 48 k5 s2 conv + GDN1 at bs 256: 1.81 -> 1.62 ms, matrix pipe 80 % busy at the
//     2.10 GHz the board then clocks (PMC, profiles/r04_f32_*): the launch sits at the power ceiling of the f32 matrix pipe
//     (155.7 TFLOP/s with register-only operands, tools/micro/mfma_f32_peak.hip);
//   * per-step tap offsets / bounds come from a small table built in LDS at the start of the workgroup;
//   * conv + GDN1 in one launch where a wave holds every channel of its pixels (Cout <= 96: SC2_EPI_FUSED_GDN): the accumulators
//     are, as they stand, the operand fragments of the 1x1 GEMM over the channels;
//   * epilogues: none, GDN1 / inverse GDN1 in the reference's operation order (norm = beta + acc; y = x * (1 / norm) resp.
//     x * norm: compressai.layers.GDN1.forward), output f32 NHWC / f32 NCHW / int32 NCHW symbols round_half_even(acc - median).
#include <stdlib.h>

#include <type_traits>

#include "sc2_common.h"

namespace {

struct F32Args {
    const float *__restrict__ x;       // f32 NHWC [N, H, W, Cin] (Cin % 4 == 0)
    const float *__restrict__ w;       // fragment-major weights
    const float *__restrict__ ep_x;    // f32 NHWC [N, OH, OW, Cout] (GDN operand) or null
    const float *__restrict__ ep_beta; // f32 [Cout]: beta (GDN) / medians (symbols) or null
    void *__restrict__ y;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
    int a_op, epilogue, out_format;
    int n_steps;                       // K_pad / 16
    long long M;                       // N * OH * OW
    unsigned x_bytes;                  // size of x (< 2 GB: the activation loads go through a buffer descriptor)
    unsigned y_bytes;                  // size of y when it is < 4 GB (the persistent first stage stores through a descriptor), else 0
    unsigned w_bytes;                  // size of w
    unsigned ring_off;                 // LDS offset of the weight ring (after the tap table, 1 KB aligned)
    int skip_j3;                       // Cin == 4 carrying 3 real channels: every fourth k is a zero channel times a zero weight
    int planar;                        // with skip_j3: x is f32 NCHW [N, 3, H, W] (the reference's input layout), read in place
};

typedef __attribute__((ext_vector_type(4))) float f4_t;

#ifndef SC2_F32_SYNC
#define SC2_F32_SYNC 1   // a workgroup barrier every two k-steps: the four waves read the SAME weight fragments, and kept within two
#endif                   // steps of each other three of the four reads are L1 hits
#ifndef SC2_F32_WAVES
#define SC2_F32_WAVES 4  // waves per SIMD the register allocation must allow for the narrow tiles (NT * MT <= 6)
#endif
#ifndef SC2_F32_MT4
#define SC2_F32_MT4 0   // experiment: four pixel tiles per wave for the 48-channel chunk (twice the MFMAs per weight fragment)
#endif

typedef __attribute__((address_space(3))) void *f32_lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t f32_rsrc_t;
__device__ __forceinline__ f32_rsrc_t f32_make_rsrc(const float *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4_t f32_buf_load16(f32_rsrc_t r, uint32_t voff) {   // out of range: zeros
    return __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void f32_buf_load_lds16(f32_rsrc_t r, uint32_t lds_addr, uint32_t voff) {   // lane l -> LDS lds_addr + 16 l
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (f32_lds_ptr_t)(uintptr_t)lds_addr, 16, (int)voff, 0, 0, 0);
}
#else   // host pass: stand-ins
typedef int f32_rsrc_t;
__device__ __forceinline__ f32_rsrc_t f32_make_rsrc(const float *, uint32_t) { return 0; }
__device__ __forceinline__ f4_t f32_buf_load16(f32_rsrc_t, uint32_t) { return f4_t{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void f32_buf_load_lds16(f32_rsrc_t, uint32_t, uint32_t) {}
#endif
// LDS reads of the k loop are inline asm (and its barriers raw s_barrier): hipcc knows that a direct-to-LDS load writes LDS and
// drains the vector-memory counter in front of every LDS access it can see -- the operand prefetch with it (conv2_gdn48.hip).
__device__ __forceinline__ f4_t f32_lds_read16(uint32_t addr) {
    f4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// The k loop issues its vector-memory instructions itself (asm) and counts them itself: hipcc's counter model put vmcnt(0) /
// vmcnt(1) in front of address arithmetic that reuses a dead operand register and in front of every LDS access after a
// direct-to-LDS load it knows about -- the two-step prefetch was gone again.  ("s_nop 4": the descriptor may have been written by
// a VALU instruction just before, e.g. v_readlane of a spilled SGPR; a VMEM instruction needs 5 wait states behind that.)
typedef int f32_desc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32_desc_t f32_make_desc(const void *base, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    return f32_desc_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void f32_astore16(const f4_t &v, f32_desc_t r, uint32_t voff) {   // out of range: dropped by the hardware
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ void f32_aload16(f4_t &d, f32_desc_t r, uint32_t voff) {   // out of range: zeros
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen ; wfrag" : "=&v"(d) : "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ void f32_dma16(f32_desc_t r, uint32_t lds_addr, uint32_t voff) {   // lane l -> LDS lds_addr + 16 l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r) : "memory", "m0");
}
__device__ __forceinline__ void f32_aload4(float &d, f32_desc_t r, uint32_t voff, uint32_t soff) {   // one float of a channel plane
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen ; wfrag" : "=&v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void f32_tie(float &v) { asm volatile("; landed %0" : "+v"(v)::"memory"); }
template <int N>
__device__ __forceinline__ void f32_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void f32_tie(f4_t &v) { asm volatile("; landed %0" : "+v"(v)::"memory"); }   // (tools/audit_vmcnt.py reads it)
// The wait names the registers it is for: an asm statement without a data dependence does not hold the VALU instructions that
// consume the read's result behind it (the first build computed one row tile's addresses from a tap entry still in flight).
__device__ __forceinline__ void f32_lds_wait(f4_t &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void f32_lds_wait(int2 &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ int2 f32_lds_read8(uint32_t addr) {
    int2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

#ifndef SC2_F32_WAVES12
#define SC2_F32_WAVES12 3
#endif
template <int NT, int MT, bool FUSED>
__global__ __launch_bounds__(256, (NT * MT <= 6 ? SC2_F32_WAVES : SC2_F32_WAVES12)) void conv_f32_kernel(F32Args p) {
    extern __shared__ int2 ktab[];     // [n_steps * 4]: {byte offset of the lane's 4 k inside the window, kh | kw << 16}
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int e = tid; e < p.n_steps * 4; e += 256) {
        const int k0 = (e >> 2) * 16 + (e & 3) * 4;
        const int tap = k0 / p.Cin, ci = k0 - tap * p.Cin;
        int2 v;
        if (tap < p.KH * p.KW) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            v.x = p.planar ? (kh * p.W + kw) * 4 : ((kh * p.W + kw) * p.Cin + ci) * 4;      // BYTE offset inside the window
            v.y = kh | (kw << 16);
        } else {            // K padding: never in bounds (its weights are zero too)
            v.x = 0;
            v.y = 0x7FFF | (0x7FFF << 16);
        }
        ktab[e] = v;
    }
    __syncthreads();

    const long long m_base = ((long long)blockIdx.x * 4 + wave) * (MT * 16);
    const int chunk = blockIdx.y;                                  // NT * 16 output channels per chunk
    // the lane's A rows: pixel m_base + mt * 16 + r
    // (byte offsets modulo 2^32: a window that starts above / left of the image has a negative base, but base + tap offset of
    //  every tap that is INSIDE the image is a plain offset below x_bytes < 2^31)
    uint32_t a_base[MT];
    int ih0[MT], iw0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long m = m_base + mt * 16 + r;
        if (m < p.M) {                             // (M < 2^31, checked by the host: 32-bit divisions)
            const uint32_t ohw_u = (uint32_t)(p.OH * p.OW), n = (uint32_t)m / ohw_u;
            const int rem = (int)((uint32_t)m - n * ohw_u);
            const int oh = (int)((uint32_t)rem / (uint32_t)p.OW), ow = rem - oh * p.OW;
            ih0[mt] = oh * p.stride - p.pad;
            iw0[mt] = ow * p.stride - p.pad;
            a_base[mt] = p.planar ? (uint32_t)(((((long long)n * 3 * p.H + ih0[mt]) * (long long)p.W + iw0[mt])) * 4)
                                  : (uint32_t)(((((long long)n * p.H + ih0[mt]) * (long long)p.W + iw0[mt]) * p.Cin) * 4);
        } else {
            ih0[mt] = iw0[mt] = -(1 << 20);       // every tap out of bounds: zeros
            a_base[mt] = 0;
        }
    }

    f4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f4_t{0.f, 0.f, 0.f, 0.f};

    // The k loop.  Activations: one 16-byte load per lane, row tile and step through a buffer descriptor (a tap outside the image is
    // an out-of-range offset = zeros from the hardware), one step ahead in registers.  Weights: the four waves of the workgroup
    // consume the SAME fragment stream, so it goes through LDS once per workgroup -- groups of four steps, wave w fetching step w
    // of the group direct-to-LDS one group ahead into a two-deep ring, one barrier per group.  (Before: every wave loaded its own
    // copy, 11.5 GB per launch of the 96 -> 48 conv through the 32 KB L1s beside 7.7 GB of activations; an L1 holds less than the
    // loads its 16 waves keep in flight, and prefetching FURTHER ahead made the launch slower.)
    const f32_desc_t rs_x = f32_make_desc(p.x, p.x_bytes);
    const f32_desc_t rs_w = f32_make_desc(p.w, p.w_bytes);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(f32_lds_ptr_t)ktab;
    const uint32_t ring = lds_base + p.ring_off;
    constexpr uint32_t GROUP_BYTES = 4u * NT * 1024u;
    const uint32_t w_chunk = (uint32_t)chunk * (uint32_t)p.n_steps * (NT * 1024u) + (uint32_t)lane * 16u;
    const int n = p.n_steps, n_groups = (n + 3) >> 2;
    auto fetch_group = [&](int g) {          // this wave's step of group g (past the end: out of range = zeros, never consumed)
        const int st = g * 4 + wave;
        const uint32_t src = st < n ? w_chunk + (uint32_t)st * (NT * 1024u) : 0x80000000u;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + (uint32_t)(g & 1) * GROUP_BYTES + (uint32_t)wave * (NT * 1024u));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) f32_dma16(rs_w, dst + nt * 1024u, src + nt * 1024u);
    };
    // an operand set of one step: MT pixel quads (NHWC input), or MT x 3 single floats out of the three channel planes (the RGB
    // image read in the reference's NCHW layout: no transposed copy)
    struct SetV { f4_t v[MT]; };
    struct SetP { float c[MT][3]; };
    const uint32_t plane_bytes = (uint32_t)(p.H * p.W) * 4u;
    auto load_a = [&](int2 t, auto &a) {
        constexpr bool PL = std::is_same<std::remove_reference_t<decltype(a)>, SetP>::value;
        const int kh = t.y & 0xFFFF, kw = t.y >> 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool ok = ((unsigned)(ih0[mt] + kh) < (unsigned)p.H) & ((unsigned)(iw0[mt] + kw) < (unsigned)p.W);
            const uint32_t vo = ok ? a_base[mt] + (uint32_t)t.x : 0x80000000u;
            if constexpr (PL) {
                f32_aload4(a.c[mt][0], rs_x, vo, 0u);
                f32_aload4(a.c[mt][1], rs_x, vo, plane_bytes);
                f32_aload4(a.c[mt][2], rs_x, vo, 2u * plane_bytes);
            } else {
                f32_aload16(a.v[mt], rs_x, vo);                // raw: |x| / x^2 where it is consumed
            }
        }
    };
    auto tie_a = [&](auto &a) {
        constexpr bool PL = std::is_same<std::remove_reference_t<decltype(a)>, SetP>::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (PL) {
                f32_tie(a.c[mt][0]);
                f32_tie(a.c[mt][1]);
                f32_tie(a.c[mt][2]);
            } else {
                f32_tie(a.v[mt]);
            }
        }
    };
    auto mma = [&](auto aop_c, const auto &a, const f4_t (&b)[NT]) {
        constexpr bool PL = std::is_same<std::remove_cv_t<std::remove_reference_t<decltype(a)>>, SetP>::value;
        constexpr int AOP = decltype(aop_c)::value & 0xFF, JN = ((decltype(aop_c)::value >> 8) & 1) ? 3 : 4;   // (bit 8: skip_j3)
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float ar;
                if constexpr (PL) ar = a.c[mt][j < 3 ? j : 0];
                else ar = a.v[mt][j];
                const float av = AOP == SC2_AOP_ABS ? fabsf(ar) : (AOP == SC2_AOP_SQUARE ? ar * ar : ar);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt][j], av, acc[mt][nt], 0, 0, 0);
            }
    };
    auto k_loop = [&](auto aop_c) {
        const uint32_t kt = lds_base + (uint32_t)q * 8u;       // tap table entry of (step, this lane's quarter): + 32 per step
        const uint32_t last = (uint32_t)(n - 1) * 32u;
        auto entry = [&](int s) { const uint32_t o = (uint32_t)s * 32u; return f32_lds_read8(kt + (o < last ? o : last)); };
        // activations S - 1 steps ahead in S register sets (a[s % S] holds step s): two ahead for the narrow tiles, one for the
        // 96-channel tile (its accumulators leave no room for a third set at three waves per SIMD).  The weight fetch of the
        // next group is issued in front of its step's activation loads in the in-order counter.
        constexpr int S = NT * MT > 6 ? 2 : 3;
        constexpr bool PL = (decltype(aop_c)::value >> 9) & 1;
        constexpr int LPS = PL ? 3 * MT : MT;                  // vector-memory instructions per step of activations
        std::conditional_t<PL, SetP, SetV> a[S];
        fetch_group(0);
        int2 t = entry(0);
        f32_lds_wait(t);
        load_a(t, a[0]);
        if (S == 3) {
            t = entry(1);
            f32_lds_wait(t);
            load_a(t, a[1]);                                   // past the end: the last step again, unused
        }
        int2 t_nxt = entry(S - 1);
        f32_lds_wait(t_nxt);
        f32_vm_wait<(S - 1) * LPS>();                          // the first group's fragments
        // One step: wait for the activations of step s, read the weights of step s and the tap entry of step s + 3 from LDS,
        // issue the next group's weight fetch (first step of a group; past the end it fetches zeros nobody reads: the counter
        // arithmetic stays the same) and the activations of step s + S - 1, multiply.
        auto step = [&](int s, auto i_c, auto ph_c, uint32_t gbuf, int g) {
            constexpr int I = decltype(i_c)::value, PH = decltype(ph_c)::value % S;
            f32_vm_wait<(S - 2) * LPS + (S == 3 && I == 1 ? NT : 0)>();   // in order: a[PH] (and everything older) has landed
            tie_a(a[PH]);
            f4_t b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = f32_lds_read16(gbuf + (uint32_t)(I * NT + nt) * 1024u + (uint32_t)lane * 16u);
            int2 t_cur = t_nxt;                                // (table entry of step s + S - 1, read one step ago)
            t_nxt = entry(s + S);
            if (I == 0) fetch_group(g + 1);
            load_a(t_cur, a[(PH + S - 1) % S]);
            f32_lds_wait(t_nxt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) f32_lds_wait(b[nt]);
            __builtin_amdgcn_sched_barrier(0);
            mma(aop_c, a[PH], b);
            __builtin_amdgcn_sched_barrier(0);
        };
        // Twelve steps = three groups = a whole number of turns of the activation sets: step s0 + K has compile-time position
        // K % 4 in its group and set K % S.  Whole turns first, then the rest as ONE chain that leaves at the first step past the end (a flat
        // control flow: the copy audit of tools/audit_vmcnt.py follows it).
        int s0 = 0;
        auto blk = [&](auto k_c) {
            constexpr int K = decltype(k_c)::value;
            const int g = (s0 + K) >> 2;
            // first step of a group: every wave's share of the group has landed (it is older than activation loads already
            // consumed); the barrier makes them visible to all four waves and frees the other half of the ring
            if (K % 4 == 0) __builtin_amdgcn_s_barrier();
            step(s0 + K, std::integral_constant<int, K % 4>{}, std::integral_constant<int, K>{},
                 ring + (uint32_t)(g & 1) * GROUP_BYTES, g);
        };
        auto turn = [&](auto... k_c) { (blk(k_c), ...); };
        auto rest = [&](auto... k_c) { (void)((s0 + decltype(k_c)::value < n && (blk(k_c), true)) && ...); };
#define SC2_F32_K12(f)                                                                                                             \
    f(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{},                        \
      std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{},                        \
      std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{},                        \
      std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{})
        for (; s0 + 12 <= n; s0 += 12) SC2_F32_K12(turn);
        SC2_F32_K12(rest);
#undef SC2_F32_K12
        f32_vm_wait<0>();                                      // (the unused loads past the end)
#pragma unroll
        for (int i = 0; i < S; ++i) tie_a(a[i]);
    };
    if (p.planar) k_loop(std::integral_constant<int, SC2_AOP_NONE | 0x300>{});
    else if (p.skip_j3) k_loop(std::integral_constant<int, SC2_AOP_NONE | 0x100>{});
    else if (p.a_op == SC2_AOP_ABS) k_loop(std::integral_constant<int, SC2_AOP_ABS>{});
    else if (p.a_op == SC2_AOP_SQUARE) k_loop(std::integral_constant<int, SC2_AOP_SQUARE>{});
    else k_loop(std::integral_constant<int, SC2_AOP_NONE>{});

    // conv FOLLOWED BY GDN1 in the same launch (SC2_EPI_FUSED_GDN / _IGDN; one channel chunk = every channel of a pixel in this
    // wave).  With the weights as the first MFMA operand a lane's accumulators acc[mt][s] ARE the second-operand fragment of
    // k-step s of the 1x1 GEMM over the channels (pixel r, channels 16 s + 4 q + j): norm = gamma |x| consumes them in place --
    // no LDS, no HBM round trip of the f32 map (1.2 GB each way for conv0 at bs 256).  Same k order as the separate GDN launch:
    // bit-identical to it.
    f4_t nrm[FUSED ? MT : 1][FUSED ? NT : 1];     // (a compile-time property: as a runtime case it cost every launch 110 registers)
    constexpr bool fused_gdn = FUSED;
    if (FUSED) {
        const f4_t *gf = reinterpret_cast<const f4_t *>(p.ep_x) + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) nrm[mt][nt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f4_t g[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) g[nt] = gf[(s * NT + nt) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        nrm[FUSED ? mt : 0][FUSED ? nt : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[nt][j], fabsf(acc[mt][s][j]), nrm[FUSED ? mt : 0][FUSED ? nt : 0], 0, 0, 0);
        }
    }

    // epilogue.  The WEIGHTS are the first MFMA operand, so the result tile is [channel][pixel]: acc[mt][nt][i] = output
    // (pixel m_base + mt * 16 + r, channel (chunk * NT + nt) * 16 + 4 q + i) -- a lane holds FOUR CONSECUTIVE CHANNELS of one
    // pixel: one 16-byte access per tile for NHWC tensors (ep_x, y), and for NCHW outputs the 16 lanes of a quarter write 16
    // consecutive pixels of a channel plane.
    const long long ohw = (long long)p.OH * p.OW;
    const bool vec4 = (p.Cout & 3) == 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long m = m_base + mt * 16 + r;
        if (m >= p.M) continue;
        const long long n_img = (uint32_t)m / (uint32_t)ohw;
        const long long pix = m - n_img * ohw;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c0 = (chunk * NT + nt) * 16 + 4 * q;
            if (c0 >= p.Cout) continue;
            float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
            float bc[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.ep_beta) {
                if (vec4) {
                    const f4_t b = *reinterpret_cast<const f4_t *>(p.ep_beta + c0);
                    bc[0] = b.x; bc[1] = b.y; bc[2] = b.z; bc[3] = b.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bc[i] = c0 + i < p.Cout ? p.ep_beta[c0 + i] : 0.f;
                }
            }
            if (fused_gdn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float norm = nrm[FUSED ? mt : 0][FUSED ? nt : 0][i] + bc[i];
                    if (p.epilogue == SC2_EPI_FUSED_GDN) norm = 1.0f / norm;
                    v[i] = v[i] * norm;
                }
            } else if (p.epilogue == SC2_EPI_GDN || p.epilogue == SC2_EPI_IGDN) {
                float xv[4];
                if (vec4) {
                    const f4_t t = *reinterpret_cast<const f4_t *>(p.ep_x + m * p.Cout + c0);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xv[i] = c0 + i < p.Cout ? p.ep_x[m * p.Cout + c0 + i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float norm = v[i] + bc[i];                 // conv2d(|x|, gamma, beta): the bias joins the finished sum
                    if (p.epilogue == SC2_EPI_GDN) norm = 1.0f / norm;   // IEEE division, then one multiply, as GDN1.forward
                    v[i] = xv[i] * norm;
                }
            } else if (p.epilogue == SC2_EPI_BIAS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += bc[i];
            }
            if (p.out_format == SC2_OUT_F32_NHWC) {
                float *dst = static_cast<float *>(p.y) + m * p.Cout + c0;
                if (vec4) {
                    *reinterpret_cast<f4_t *>(dst) = f4_t{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c0 + i < p.Cout) dst[i] = v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (c0 + i >= p.Cout) continue;
                    const long long o = (n_img * p.Cout + c0 + i) * ohw + pix;
                    if (p.out_format == SC2_OUT_F32_NCHW) static_cast<float *>(p.y)[o] = v[i];
                    else static_cast<int32_t *>(p.y)[o] = (int32_t)rintf(v[i] - bc[i]);   // symbols: bc = the channel's median
                }
            }
        }
    }
}

// The first encoder stage of the reference-precision path as a PERSISTENT kernel: Conv2d(3 -> 96, k5, s2, p2) on the RGB planes +
// GDN1(96), f32 NHWC out.  K is 7 steps (+ 6 of the norm GEMM): in the tile-per-workgroup form above a wave lives for 13 steps,
// and prologue, first-load latency and epilogue of every 128 pixels leave the matrix pipe 35 % idle (PMC).  Here a workgroup
// loads the conv's fragments (42 KB), gamma's (36 KB) and beta into LDS ONCE and its four waves walk 32-pixel tiles: the next
// tile's 42 plane loads are issued when the conv phase has consumed the operand registers and land behind the 288 MFMAs of the
// norm phase.  Same products in the same order as the tile form: bit-identical (`test_conv_f32_persist_equals_tile_form`).
constexpr int P0_STEPS = 7, P0_NT = 6, P0_MT = 2;
constexpr int P0_W_FRAGS = P0_STEPS * P0_NT * 64, P0_G_FRAGS = P0_NT * P0_NT * 64;   // 16-byte fragments

template <bool INVERSE>
__global__ __launch_bounds__(256, 2) void conv0_gdn_f32_persist_kernel(const F32Args p) {
    extern __shared__ f4_t pl[];                     // [P0_W_FRAGS] conv, [P0_G_FRAGS] gamma, [24] beta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    {
        const f4_t *wsrc = reinterpret_cast<const f4_t *>(p.w), *gsrc = reinterpret_cast<const f4_t *>(p.ep_x);
        for (int e = tid; e < P0_W_FRAGS; e += 256) pl[e] = wsrc[e];
        for (int e = tid; e < P0_G_FRAGS; e += 256) pl[P0_W_FRAGS + e] = gsrc[e];
        if (tid < 24) pl[P0_W_FRAGS + P0_G_FRAGS + tid] = reinterpret_cast<const f4_t *>(p.ep_beta)[tid];
    }
    __syncthreads();
    const f32_desc_t rs_x = f32_make_desc(p.x, p.x_bytes);
    const f32_desc_t rs_y = f32_make_desc(p.y, p.y_bytes);
    const uint32_t plane_bytes = (uint32_t)(p.H * p.W) * 4u;
    // this lane's tap of every step: k-quarter q of step s is tap 4 s + q (25 .. 27: padding, never in bounds)
    uint32_t tap_off[P0_STEPS];
    int tap_hw[P0_STEPS];
#pragma unroll
    for (int s = 0; s < P0_STEPS; ++s) {
        const int tap = 4 * s + q, kh = tap / 5, kw = tap - kh * 5;
        tap_off[s] = (uint32_t)((kh * p.W + kw) * 4);
        tap_hw[s] = tap < 25 ? (kh | (kw << 16)) : (0x7FFF | (0x7FFF << 16));
    }
    const uint32_t ohw = (uint32_t)(p.OH * p.OW), M = (uint32_t)p.M;
    const int n_tiles = (int)((p.M + 31) / 32), tile_step = gridDim.x * 4;
    float a[P0_STEPS][P0_MT][3];
    auto fetch = [&](int tile) {                     // 42 plane loads of this lane's two pixels (tile past the end: zeros)
#pragma unroll
        for (int mt = 0; mt < P0_MT; ++mt) {
            const uint32_t m = (uint32_t)tile * 32u + (uint32_t)(mt * 16 + r);
            const bool live = tile < n_tiles && m < M;
            const uint32_t mm = live ? m : 0u;
            const uint32_t n = mm / ohw, rem = mm - n * ohw, oh = rem / (uint32_t)p.OW, ow = rem - oh * (uint32_t)p.OW;
            const int ih0 = live ? (int)oh * 2 - 2 : -(1 << 20), iw0 = (int)ow * 2 - 2;
            const uint32_t base = (uint32_t)((((long long)n * 3 * p.H + ih0) * (long long)p.W + iw0) * 4);
#pragma unroll
            for (int s = 0; s < P0_STEPS; ++s) {
                const int kh = tap_hw[s] & 0xFFFF, kw = tap_hw[s] >> 16;
                const bool ok = ((unsigned)(ih0 + kh) < (unsigned)p.H) & ((unsigned)(iw0 + kw) < (unsigned)p.W);
                const uint32_t vo = ok ? base + tap_off[s] : 0x80000000u;
                f32_aload4(a[s][mt][0], rs_x, vo, 0u);
                f32_aload4(a[s][mt][1], rs_x, vo, plane_bytes);
                f32_aload4(a[s][mt][2], rs_x, vo, 2u * plane_bytes);
            }
        }
    };
    int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;                     // (no barrier below)
    fetch(tile);
    f32_vm_wait<0>();                                // (the first tile has no stores behind its loads: the counted wait below would let 12 LOADS stay out)
    for (; tile < n_tiles; tile += tile_step) {
        f32_vm_wait<P0_MT * P0_NT>();                // in order: everything but the previous tile's 12 stores, issued behind the loads
#pragma unroll
        for (int s = 0; s < P0_STEPS; ++s)
#pragma unroll
            for (int mt = 0; mt < P0_MT; ++mt) {
                f32_tie(a[s][mt][0]);
                f32_tie(a[s][mt][1]);
                f32_tie(a[s][mt][2]);
            }
        f4_t acc[P0_MT][P0_NT], nrm[P0_MT][P0_NT];
#pragma unroll
        for (int mt = 0; mt < P0_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < P0_NT; ++nt) acc[mt][nt] = nrm[mt][nt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < P0_STEPS; ++s) {
            f4_t b[P0_NT];
#pragma unroll
            for (int nt = 0; nt < P0_NT; ++nt) b[nt] = pl[(s * P0_NT + nt) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int mt = 0; mt < P0_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < P0_NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt][j], a[s][mt][j], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(tile + tile_step);                     // the operand registers are free: the next tile's loads, behind the norm phase
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < P0_NT; ++s) {
            f4_t g[P0_NT];
#pragma unroll
            for (int nt = 0; nt < P0_NT; ++nt) g[nt] = pl[P0_W_FRAGS + (s * P0_NT + nt) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < P0_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < P0_NT; ++nt)
                        nrm[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[nt][j], fabsf(acc[mt][s][j]), nrm[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < P0_MT; ++mt) {
            const uint32_t m = (uint32_t)tile * 32u + (uint32_t)(mt * 16 + r);
#pragma unroll
            for (int nt = 0; nt < P0_NT; ++nt) {
                const f4_t bc = pl[P0_W_FRAGS + P0_G_FRAGS + nt * 4 + q];
                f4_t v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float norm = nrm[mt][nt][i] + bc[i];
                    if (!INVERSE) norm = 1.0f / norm;          // IEEE division, then one multiply, as GDN1.forward
                    v[i] = acc[mt][nt][i] * norm;
                }
                // (round 5: EVERY tile issues its twelve stores -- rows past the end of the tensor go out of the descriptor's range and
                //  are dropped by the hardware -- so that `vmcnt(12)` at the top of the loop is exact on every path; with a store the
                //  compiler branched around, tools/audit_vmcnt.py --counts could only be told, not shown, that no iteration follows)
                f32_astore16(v, rs_y, m < M ? (uint32_t)(m * 384u + (uint32_t)(nt * 16 + 4 * q) * 4u) : 0x80000000u);
            }
        }
    }
    f32_vm_wait<0>();                                // (the loads of a tile past the end)
}

bool f32_persist0_enabled() {     // SC2_F32_PERSIST0=0: the tile form (A/B)
    return sc2_pol().f32_persist0 != 0;
}

template <int NT, int MT, bool FUSED = false>
int launch_f32(const F32Args &a, int chunks, hipStream_t s) {
    const long long tiles = (a.M + (4 * MT * 16) - 1) / (4 * MT * 16);
    F32Args b = a;
    b.ring_off = (unsigned)(((size_t)a.n_steps * 4 * sizeof(int2) + 1023) / 1024 * 1024);
    b.w_bytes = (unsigned)((size_t)chunks * a.n_steps * NT * 1024);
    const size_t lds = (size_t)b.ring_off + 2 * 4 * NT * 1024;
    hipLaunchKernelGGL((conv_f32_kernel<NT, MT, FUSED>), dim3((unsigned)tiles, (unsigned)chunks), dim3(256), lds, s, b);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

// f32 NCHW -> f32 NHWC with the channel count padded (zeros): the layout the f32 conv reads.
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int HW,
                                                               int Cpad, long long total_pix) {
    const int c0 = blockIdx.y * 4;
    for (long long gp = (long long)blockIdx.x * 256 + threadIdx.x; gp < total_pix; gp += (long long)gridDim.x * 256) {
        const long long n = gp / HW;
        const int pix = (int)(gp - n * HW);
        f4_t v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c0 + j < C ? x[(n * C + c0 + j) * HW + pix] : 0.f;
        *reinterpret_cast<f4_t *>(y + gp * Cpad + c0) = v;
    }
}

}  // namespace

extern "C" int sc2_conv_f32_chunk_channels(int Cout) { return Cout <= 32 ? 32 : (Cout <= 48 ? 48 : 96); }

extern "C" int sc2_nchw_f32_to_nhwc_f32(const float *x, float *y, int N, int C, int H, int W, int Cpad, void *stream) {
    SC2_REQUIRE(x && y, SC2_ERR_INVALID_ARG, "nchw_to_nhwc_f32: null argument");
    SC2_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 4 == 0, SC2_ERR_INVALID_ARG,
                "nchw_to_nhwc_f32: bad dims N=%d C=%d H=%d W=%d Cpad=%d", N, C, H, W, Cpad);
    const long long total = (long long)N * H * W;
    const int gx = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(gx, Cpad / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, C,
                       H * W, Cpad, total);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_conv2d_f32_fwd(const sc2_conv_desc *d, const float *x, const float *w_frag, void *y, const float *ep_x,
                                  const float *ep_beta, void *stream) {
    SC2_REQUIRE(d && x && w_frag && y, SC2_ERR_INVALID_ARG, "conv2d_f32: null argument");
    SC2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 4 == 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0,
                SC2_ERR_INVALID_ARG, "conv2d_f32: bad dims (Cin must be a multiple of 4)");
    SC2_REQUIRE(d->stride_h == d->stride_w && d->pad_h == d->pad_w && d->stride_h > 0 && d->pad_h >= 0, SC2_ERR_UNSUPPORTED,
                "conv2d_f32: square stride / padding only");
    SC2_REQUIRE(d->OH == (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 && d->OW == (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1,
                SC2_ERR_INVALID_ARG, "conv2d_f32: OH / OW do not match the geometry");
    SC2_REQUIRE(d->out_H == 0, SC2_ERR_UNSUPPORTED, "conv2d_f32: no output scatter");
    SC2_REQUIRE(d->a_op == SC2_AOP_NONE || d->a_op == SC2_AOP_ABS || d->a_op == SC2_AOP_SQUARE, SC2_ERR_INVALID_ARG, "conv2d_f32: a_op");
    const bool fused = d->epilogue == SC2_EPI_FUSED_GDN || d->epilogue == SC2_EPI_FUSED_IGDN;
    SC2_REQUIRE(d->epilogue == SC2_EPI_NONE || d->epilogue == SC2_EPI_GDN || d->epilogue == SC2_EPI_IGDN || d->epilogue == SC2_EPI_BIAS || fused,
                SC2_ERR_UNSUPPORTED, "conv2d_f32: epilogue %d", d->epilogue);
    SC2_REQUIRE(!fused || (d->Cout <= 96 && ep_x && ep_beta), SC2_ERR_UNSUPPORTED,
                "conv2d_f32: the fused GDN needs every channel of a pixel in one chunk (Cout <= 96), gamma fragments and beta");
    // the fused norm GEMM walks chunk / 16 k-steps of gamma fragments; gamma is packed with ceil(Cout / 16) of them (a 1x1 weight
    // of K = Cout): a narrower Cout (49 .. 80, or <= 16) would read up to 12 KB past the tensor
    SC2_REQUIRE(!fused || (d->Cout + 15) / 16 * 16 == sc2_conv_f32_chunk_channels(d->Cout), SC2_ERR_UNSUPPORTED,
                "conv2d_f32: the fused GDN needs ceil(Cout / 16) * 16 == the chunk width (%d channels: chunk %d); run conv and GDN1 as two launches",
                d->Cout, sc2_conv_f32_chunk_channels(d->Cout));
    SC2_REQUIRE(d->out_format == SC2_OUT_F32_NHWC || d->out_format == SC2_OUT_F32_NCHW || d->out_format == SC2_OUT_I32_NCHW_SYM,
                SC2_ERR_UNSUPPORTED, "conv2d_f32: out_format %d", d->out_format);
    const bool gdn = d->epilogue == SC2_EPI_GDN || d->epilogue == SC2_EPI_IGDN;
    SC2_REQUIRE(!gdn || (ep_x && ep_beta), SC2_ERR_INVALID_ARG, "conv2d_f32: GDN epilogue needs ep_x and ep_beta");
    SC2_REQUIRE((d->epilogue != SC2_EPI_BIAS && d->out_format != SC2_OUT_I32_NCHW_SYM) || ep_beta, SC2_ERR_INVALID_ARG,
                "conv2d_f32: ep_beta (bias / medians) missing");
    SC2_REQUIRE(d->out_format != SC2_OUT_I32_NCHW_SYM || d->epilogue == SC2_EPI_NONE, SC2_ERR_INVALID_ARG,
                "conv2d_f32: symbols come straight from the accumulators (epilogue NONE)");
    F32Args a;
    a.x = x; a.w = w_frag; a.ep_x = ep_x; a.ep_beta = ep_beta; a.y = y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride_h; a.pad = d->pad_h; a.OH = d->OH; a.OW = d->OW;
    a.a_op = d->a_op; a.epilogue = d->epilogue; a.out_format = d->out_format;
    a.n_steps = (d->KH * d->KW * d->Cin + 15) / 16;
    // Kpad (the bf16 kernels' weight pitch) carries the REAL channel count here: 3 of Cin == 4 means the fourth channel of x and the
    // weights of k % 4 == 3 are zero (hip.nchw_f32_to_nhwc_f32 / hip.pack_conv_f32 make them so) and their products are skipped
    a.skip_j3 = (d->Cin == 4 && d->Kpad == 3 && d->a_op == SC2_AOP_NONE) ? 1 : 0;
    // k_order (ignored otherwise) = 1 with it: x is the f32 NCHW image [N, 3, H, W] itself, the three planes read in place
    a.planar = d->k_order == 1 ? 1 : 0;
    SC2_REQUIRE(!a.planar || a.skip_j3, SC2_ERR_UNSUPPORTED, "conv2d_f32: the NCHW input form needs Cin == 4 with Kpad == 3 and a_op NONE");
    a.M = (long long)d->N * d->OH * d->OW;
    {
        const long long xb = (long long)d->N * d->H * d->W * (d->k_order == 1 ? 3 : d->Cin) * 4;
        SC2_REQUIRE(xb < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv2d_f32: input of %lld bytes exceeds 2 GB", xb);
        a.x_bytes = (unsigned)xb;
        a.y_bytes = 0u;
    }
    SC2_REQUIRE((long long)d->N * d->H * d->W * d->Cin < (1ll << 31) && a.M * d->Cout < (1ll << 33), SC2_ERR_UNSUPPORTED,
                "conv2d_f32: tensor too large for this kernel's index arithmetic");
    SC2_REQUIRE((size_t)a.n_steps * 32 <= 64 * 1024, SC2_ERR_UNSUPPORTED, "conv2d_f32: K too long for the tap table");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cc = sc2_conv_f32_chunk_channels(d->Cout);
    const int chunks = (d->Cout + cc - 1) / cc;
    // (measured: four row tiles per wave for the narrow chunks -- twice the MFMAs per operand load -- ran the 96 -> 48 k5 s2 conv
    //  in 2.89 ms instead of 2.38 at bs 256: fewer, fatter waves hide less of the operand latency; two row tiles everywhere)
    constexpr int MT48 = SC2_F32_MT4 ? 4 : 2;
    if (fused && a.planar && d->Cout == 96 && d->KH == 5 && d->KW == 5 && d->stride_h == 2 && d->pad_h == 2 && d->out_format == SC2_OUT_F32_NHWC &&
        f32_persist0_enabled() && a.M * 96LL * 4LL < 0xFFFFFF00LL) {
        a.y_bytes = (unsigned)(a.M * 96LL * 4LL);
        static int n_cu_dev[SC2_MAX_DEVICES] = {};     // per device, as the function attributes below
        int &n_cu = n_cu_dev[sc2_device_slot()];
        if (n_cu == 0) {
            int dev = 0, v = 0;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        }
        const size_t lds = (size_t)(P0_W_FRAGS + P0_G_FRAGS + 24) * 16;
        const long long tiles = (a.M + 31) / 32;
        int grid = n_cu * 2;
        if ((long long)grid * 4 > tiles) grid = (int)((tiles + 3) / 4);
        static bool attr_set_dev[SC2_MAX_DEVICES] = {};
        bool &attr_set = attr_set_dev[sc2_device_slot()];
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn_f32_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv0_gdn_f32_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (d->epilogue == SC2_EPI_FUSED_IGDN) hipLaunchKernelGGL(conv0_gdn_f32_persist_kernel<true>, dim3(grid), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(conv0_gdn_f32_persist_kernel<false>, dim3(grid), dim3(256), lds, s, a);
        SC2_CHECK_LAUNCH();
        return SC2_OK;
    }
    if (fused) {
        if (cc == 32) return launch_f32<2, 2, true>(a, chunks, s);
        if (cc == 48) return launch_f32<3, MT48, true>(a, chunks, s);
        return launch_f32<6, 2, true>(a, chunks, s);
    }
    if (cc == 32) return launch_f32<2, 2>(a, chunks, s);
    if (cc == 48) return launch_f32<3, MT48>(a, chunks, s);
    return launch_f32<6, 2>(a, chunks, s);
}
