// Persistent form of the 256 x 256 big-tile kernel for the two MFMA-bound decoder layers of the FP / SHP / MSHP
// bottlenecks (gfx950):
//     dec.conv2 (512 -> 256, k2, p0) + inverse GDN1(256)   (sc2bench/models/layer.py:489-491)
//     dec.conv4 (256 -> 256, k2, p1)                        (layer.py:492-493)
//
// The K loop is the one of conv_igemm8_kernel (conv_igemm_impl.h: 8 waves in two groups one barrier out of step, BK = 32
// slabs in a 4-deep direct-to-LDS ring, buffer-addressed loads).  What changes is everything AROUND it.  As one workgroup
// per tile, a CU spends per tile: ~2 us until the first slab lands, the K loop, the store epilogue (accumulators -> bf16
// LDS image -> 128 KB of global stores) and its drain -- and with one 160 KB workgroup per CU nothing overlaps any of it
// (measured, tools/attic/epi_share.sh: dec.conv2 0.915 ms in all, 0.72 - 0.77 ms with the store epilogue switched off).
// Here a workgroup walks the tiles of its XCD's share of the output and
//   * parks the finished tile's 128 KB in REGISTERS (16 x 16 bytes per thread) once the image is complete,
//   * computes the next tile's gather state and issues its first three slabs,
//   * THEN issues the 16 output stores of the finished tile: they are younger than those slabs, so the counted vmcnt
//     waits of the first K-loop iterations (+16) do not wait for store acknowledgements; by the third iteration
//     (~1 us later) they have retired,
// so the stores drain and the first slabs arrive while the matrix pipes already work on the next tile.
//
// Measured on one MI355X, bs 256, same process (tools/attic/ab_env.sh SC2_CONV_PERSIST "0 1 2 3"):
//                                     dec.conv2   dec.conv2 + IGDN256   dec.conv4
//   0  one workgroup per tile           0.92 ms        1.02 ms           0.52 ms
//   1  persistent, deferred stores      0.79           0.90              0.45
//   2  + fragment reads a phase early   0.75           0.85              0.43
//   3  ONE phase per slab (default)     0.71           0.81              0.41
// What bounds mode 3 (timing experiments with the operand loads switched off, outputs garbage): no loads at all 0.40 ms
// (2.0 PFLOP/s: the MFMA / fragment-read / barrier structure alone), pixel operand only or weight operand only 0.52 - 0.53,
// both 0.72: the 6 GB a launch moves L2 -> LDS (1 MB of pixels + 1 MB of weights per tile) arrive at ~19 TB/s, the
// direct-to-LDS gather rate of the chip (MI355X_MICROARCH.md, indexed rows: 16.8 - 18.8 TB/s from L2), and an issuing
// wave that is held back by that queue holds its barrier partner back.  Cache-policy bits on the loads change nothing.
// Fewer bytes would need the four taps to share one staged window; with the XOR-swizzled 64-byte rows the shifted reads
// then pay address arithmetic inside the load interval, which is the critical path (tried in round 1: 25 % slower).
#include "conv_igemm_impl.h"

namespace sc2conv {

// ds_read_b128 at addr + OFF (immediate): the fragment rows of a lane are 1 KB apart (16 tile rows x 64 B, same swizzle),
// so ONE address register serves all eight pixel fragments and one the four weight fragments
template <int OFF>
__device__ __forceinline__ uint4 lds_read16_imm(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// PIPE: the fragment reads of a phase are issued one phase EARLY, at the top of the previous phase's MFMA interval (second
// register sets for the pixel and weight fragments), so that a wave's load interval holds only its two direct-to-LDS loads
// and the wait for reads issued ~300 cycles before; the wait that retires slab kt + 1 moves from the last to the first
// phase of slab kt so that BOTH wave groups have retired it before either reads it (the groups run one barrier apart).
// MODE 2: ONE phase per slab -- 32 MFMAs between two barriers instead of 16 (eight pixel fragments + four weight fragments
// read and both direct-to-LDS halves issued in the one load interval): half the barriers per MFMA.
template <class C, int MODE>
__global__ __launch_bounds__(512, 2) void conv_igemm8p_kernel(const ConvArgs p, const int n_tiles, const int chunk) {
    constexpr bool PIPE = MODE == 1, ONE = MODE == 2 || MODE == 3;
    constexpr bool X32 = MODE == 3;   // TIMING EXPERIMENT (results garbage): the slab as 16 32x32x16 MFMAs on the same fragments
    constexpr int BM = C::BM, BN = C::BN;
    constexpr int MT = C::MT, NT = C::NT, S = C::STAGES, PHASES = C::PHASES;
    constexpr int A_IPW = C::A_IPW, B_IPW = C::B_IPW, L = A_IPW + B_IPW;
    constexpr int KH = C::KH, KW = C::KW, SH = C::SH, SW = C::SW, PH = C::PH, PW = C::PW, Cin = C::CIN;
    static_assert(C::STATIC && Cin % 32 == 0 && !C::PATCH3 && BN == 256 && BM == 256 && S == 4 && KH * KW <= 32, "geometry");
    constexpr int CPR = BN / 8, QPT = BM * CPR / 512;   // 16-byte chunks per tile row (32), per thread (16)
    constexpr uint32_t OOB = 0x80000000u;
    constexpr int WAIT_LOOP = PIPE ? (S - 3) * L + A_IPW : (S - 2) * L, WAIT_PEND = WAIT_LOOP + QPT;
    constexpr int WAIT_PRO = (S - 2) * L, WAIT_PRO_PEND = WAIT_PRO + QPT;
    static_assert(!PIPE || PHASES == 2, "two phases per slab");
    static_assert(WAIT_PEND < 64, "vmcnt is a 6-bit count");
    typedef ImgXor<C> Img;
    static_assert(Img::BYTES <= S * C::STAGE_BYTES, "the output image fits the ring");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;   // waves w and w + 4 sit on the same SIMD and work out of step
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int H = p.H, W = p.W;
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    const int frow = lane & 15, fq = lane >> 4;

    // descriptors: x starts at tap (0, 0) of pixel (0, 0); out-of-image lanes are sent out of range (zeros)
    const long long shift = ((long long)PH * W + PW) * Cin;
    const buf_rsrc_t rs_x = make_rsrc(p.x - shift, p.x_bytes + (uint32_t)(shift * 2));
    const buf_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    uint32_t b_vo[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j)
        b_vo[j] = (uint32_t)(((long long)((j * 8 + wave) * 16 + (lane >> 2)) * p.b_row_stride + kc * 8) * 2);
    uint32_t a_rd[MT], b_rd[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rd[i] = (uint32_t)lds_off(wm * C::WM + i * 16 + frow, fq);
#pragma unroll
    for (int j = 0; j < NT; ++j) b_rd[j] = (uint32_t)(C::A_BYTES + lds_off(wn * C::WN + j * 16 + frow, fq));
    const int KT = p.KT;
    constexpr int spt = Cin / 32;   // slabs per tap
    const bool fused = p.epi == SC2_EPI_FUSED_GDN || p.epi == SC2_EPI_FUSED_IGDN;
    uint16_t *const y = reinterpret_cast<uint16_t *>(p.y);

    // this workgroup's tiles: XCD x owns a contiguous range of the output (halo rows and the weight panel stay in ONE
    // L2), walked in order by the workgroups dispatched to it (round-robin: blockIdx & 7)
    // and cut into runs of `chunk` consecutive tiles, one run per workgroup.  chunk = the whole share of a CU: fully
    // persistent; smaller: more workgroups than CUs, which the hardware dispatcher spreads over whatever CUs are free
    // (the serial coder's 136 KB workgroups and the encoder stage of a neighbouring batch take CUs away for milliseconds:
    // with one static share per CU the launch then waits for the workgroups that could not start)
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int tq = n_tiles >> 3, tr = n_tiles & 7;
    const int t_base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_cnt = tq + (xcd < tr ? 1 : 0);
    const int t_first = wl * chunk;
    const int t_last = t_first + chunk < t_cnt ? t_first + chunk : t_cnt;

    // the previous tile's output, parked until the next tile's first slabs are issued: SIXTEEN NAMED registers quads (as
    // an array carried around the tile loop it stayed in scratch memory: 256 B / lane of scratch traffic whose reloads
    // wait vmcnt(0) in the middle of the counted DMA schedule)
    static_assert(QPT == 16, "sixteen parked chunks per thread");
#define SC2_PEND_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define SC2_PEND_DECL(r) uint4 pend##r = make_uint4(0u, 0u, 0u, 0u);
    SC2_PEND_LIST(SC2_PEND_DECL)
    bool has_pend = false;
    int pend_m0 = 0;
    // (tq: an opaque copy of tid made where it is used -- the per-chunk offsets are then recomputed per tile instead of being
    //  hoisted out of the tile loop and spilled around the K loop, whose reloads would wait vmcnt(0))
#define SC2_PEND_STORE(r)                                                                          \
    {                                                                                              \
        const int q = tq + (r) * 512;                                                              \
        const int row = q / CPR, cc = q - row * CPR;                                               \
        const int m = pend_m0 + row;                                                               \
        if (m < p.M) *reinterpret_cast<uint4 *>(y + (long long)m * BN + cc * 8) = pend##r;         \
    }
#define SC2_PEND_LOAD(r)                                                                           \
    {                                                                                              \
        const int q = tq + (r) * 512;                                                              \
        const int row = q / CPR, cc = q - row * CPR;                                               \
        pend##r = *reinterpret_cast<const uint4 *>(img + Img::off(row, cc));                       \
    }

    for (int t = t_first; t < t_last; ++t) {
        const int m0 = (t_base + t) * BM;
        // ---- gather state of this tile: per-lane byte offset of its pixel, taps that fall inside the image
        uint32_t a_vo[A_IPW], a_tapmask[A_IPW];
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int m = m0 + (j * 8 + wave) * 16 + (lane >> 2);
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int img = mm / p.OHW;
            const int rem = mm - img * p.OHW;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            a_vo[j] = (uint32_t)((((long long)img * H + oh * SH) * W + ow * SW) * Cin * 2 + kc * 16);
            uint32_t mk = 0;
#pragma unroll
            for (int tp = 0; tp < KH * KW; ++tp) {
                const int ih = oh * SH - PH + tp / KW, iw = ow * SW - PW + tp % KW;
                mk |= (ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W)) ? (1u << tp) : 0u;
            }
            a_tapmask[j] = mk;
        }
        int next_a = 0;
        auto issue_a = [&](int buf) {
            unsigned char *Ab = smem + buf * C::STAGE_BYTES;
            int tap, cb;   // scalar: (tap, channel block) of the slab
            if (p.k_slab_major) { cb = next_a / (KH * KW); tap = next_a - cb * (KH * KW); }
            else { tap = next_a / spt; cb = next_a - tap * spt; }
            const bool tap_ok = next_a < KT;   // false for the dummy slabs past KT
            const uint32_t soff = (uint32_t)(((tap / KW) * W + tap % KW) * Cin + cb * 32) * 2u;
#pragma unroll
            for (int j = 0; j < A_IPW; ++j) {
                const uint32_t vo = (tap_ok && ((a_tapmask[j] >> tap) & 1u)) ? a_vo[j] : OOB;
                buf_load_lds16(rs_x, (lds_ptr_t)(Ab + (j * 8 + wave) * 1024), vo, soff);
            }
            ++next_a;
        };
        auto issue_b = [&](int kt, int buf) {
            unsigned char *Bb = smem + buf * C::STAGE_BYTES + C::A_BYTES;
            const uint32_t soff = (uint32_t)(kt < KT ? kt : KT - 1) * (uint32_t)p.b_kt_stride * 2u;   // (slabs past KT are never read)
#pragma unroll
            for (int j = 0; j < B_IPW; ++j) buf_load_lds16(rs_w, (lds_ptr_t)(Bb + (j * 8 + wave) * 1024), b_vo[j], soff);
        };

#pragma unroll
        for (int st = 0; st < S - 1; ++st) {
            issue_a(st);
            issue_b(st, st);
        }
        // the finished tile leaves now: its stores are younger than the three slabs above
        asm volatile("" ::: "memory");
        if (has_pend) {
            int tq = tid;
            asm volatile("" : "+v"(tq));
            SC2_PEND_LIST(SC2_PEND_STORE)
        }
        asm volatile("" ::: "memory");

        // (zeroed by volatile asm: plain initialisers are hoisted above the stores, where the 64 parked registers and the
        //  128 accumulators do not fit together)
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float z0, z1, z2, z3;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
                             : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3));
                acc[i][j] = f32x4_t{z0, z1, z2, z3};
            }

        // slab 0 has landed (this wave's share); with stores pending, they and two slabs may still be in flight
        if (has_pend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PRO_PEND) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PRO) : "memory");
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

        if constexpr (PIPE) {
            // fragments: two pixel sets (av0: tile rows 0-3 of a slab, phase 0; av1: rows 4-7, phase 1) and two weight sets
            // (bvA: even slabs, bvB: odd slabs); each is read during the MFMA interval of the phase before its use
            uint4 av0[4], av1[4], bvA[NT], bvB[NT];
            static_assert(NT == 4, "four weight fragments per wave");
            const uint32_t a_rd0 = a_rd[0], b_rd0 = b_rd[0];   // a_rd[i] = a_rd0 + 1024 i, b_rd[j] = b_rd0 + 1024 j
            auto read_b = [&](uint32_t base, uint4 (&b)[NT]) {
                b[0] = lds_read16_imm<0>(base + b_rd0);
                b[1] = lds_read16_imm<1024>(base + b_rd0);
                b[2] = lds_read16_imm<2048>(base + b_rd0);
                b[3] = lds_read16_imm<3072>(base + b_rd0);
            };
            auto read_a_lo = [&](uint32_t base, uint4 (&a)[4]) {
                a[0] = lds_read16_imm<0>(base + a_rd0);
                a[1] = lds_read16_imm<1024>(base + a_rd0);
                a[2] = lds_read16_imm<2048>(base + a_rd0);
                a[3] = lds_read16_imm<3072>(base + a_rd0);
            };
            auto read_a_hi = [&](uint32_t base, uint4 (&a)[4]) {
                a[0] = lds_read16_imm<4096>(base + a_rd0);
                a[1] = lds_read16_imm<5120>(base + a_rd0);
                a[2] = lds_read16_imm<6144>(base + a_rd0);
                a[3] = lds_read16_imm<7168>(base + a_rd0);
            };
            read_b(lds_base, bvA);
            read_a_lo(lds_base, av0);
            auto mfma16 = [&](int half, const uint4 (&a)[4], const uint4 (&b)[NT]) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, b[j]), __builtin_bit_cast(bf16x8_t, a[i]), acc[4 * half + i][j], 0, 0, 0);
            };
            auto slab = [&](int kt, const uint4 (&bc)[NT], uint4 (&bn)[NT]) {
                const uint32_t sb = lds_base + (uint32_t)((kt % S) * C::STAGE_BYTES);
                const uint32_t sb1 = lds_base + (uint32_t)(((kt + 1) % S) * C::STAGE_BYTES);
                const int nbuf = (kt + S - 1) % S;
                // ---- phase 0
                issue_a(nbuf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // av0, bc
                // slab kt + 1 has landed (this wave's share); the stores of the previous tile sit between slab 2 and slab 3
                if (has_pend && kt < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PEND) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_LOOP) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                read_a_hi(sb, av1);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(0, av0, bc);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase 1
                issue_b(kt + S - 1, nbuf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // av1
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                read_b(sb1, bn);   // slab kt + 1: retired by both groups a phase ago
                read_a_lo(sb1, av0);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(1, av1, bc);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int kt = 0; kt < KT; kt += 2) {   // (KT is even: Cin % 64 == 0 with four taps)
                slab(kt, bvA, bvB);
                slab(kt + 1, bvB, bvA);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments read ahead for the slab past the end
        } else if constexpr (ONE) {
            const uint32_t a_rd0 = a_rd[0], b_rd0 = b_rd[0];
            typedef __attribute__((ext_vector_type(16))) float f32x16_t;
            [[maybe_unused]] f32x16_t acc32[4][2];
            if constexpr (X32) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc32[a][b][e] = 0.f;
            }
            for (int kt = 0; kt < KT; ++kt) {
                const uint32_t sb = lds_base + (uint32_t)((kt % S) * C::STAGE_BYTES);
                const int nbuf = (kt + S - 1) % S;
                uint4 bv[NT], av[MT];
                static_assert(NT == 4 && MT == 8, "fragment counts");
                bv[0] = lds_read16_imm<0>(sb + b_rd0); bv[1] = lds_read16_imm<1024>(sb + b_rd0);
                bv[2] = lds_read16_imm<2048>(sb + b_rd0); bv[3] = lds_read16_imm<3072>(sb + b_rd0);
                av[0] = lds_read16_imm<0>(sb + a_rd0); av[1] = lds_read16_imm<1024>(sb + a_rd0);
                av[2] = lds_read16_imm<2048>(sb + a_rd0); av[3] = lds_read16_imm<3072>(sb + a_rd0);
                av[4] = lds_read16_imm<4096>(sb + a_rd0); av[5] = lds_read16_imm<5120>(sb + a_rd0);
                av[6] = lds_read16_imm<6144>(sb + a_rd0); av[7] = lds_read16_imm<7168>(sb + a_rd0);
                issue_a(nbuf);
                issue_b(kt + S - 1, nbuf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (has_pend && kt < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PEND) : "memory");   // slab kt + 1 has landed
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_LOOP) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if constexpr (X32) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                acc32[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv[2 * b + h]),
                                                                                      __builtin_bit_cast(bf16x8_t, av[2 * a + h]), acc32[a][b], 0, 0, 0);
                } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv[j]),
                                                                            __builtin_bit_cast(bf16x8_t, av[i]), acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (X32) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int q = 4 * ((i & 1) * 2 + (j & 1));
                        acc[i][j] = f32x4_t{acc32[i >> 1][j >> 1][q], acc32[i >> 1][j >> 1][q + 1], acc32[i >> 1][j >> 1][q + 2],
                                            acc32[i >> 1][j >> 1][q + 3]};
                    }
            }
        } else {
        uint4 bv[NT];
        for (int kt = 0; kt < KT; ++kt) {
            const uint32_t sb = lds_base + (uint32_t)((kt % S) * C::STAGE_BYTES);
            const int nbuf = (kt + S - 1) % S;
            // the stores sit between slab 2 and slab 3 in issue order: the waits of iterations 0 and 1 (for slabs 1 and 2)
            // leave them in flight, from iteration 2 on (slab 3) they are older than what is waited for
            const bool pend_in_flight = has_pend && kt < 2;
#pragma unroll
            for (int ph = 0; ph < PHASES; ++ph) {
                uint4 av[4];
                if (ph == 0) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = lds_read16(sb + b_rd[j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = lds_read16(sb + a_rd[4 * ph + i]);
                if (ph == 0) issue_a(nbuf);
                if (ph == PHASES - 1) issue_b(kt + S - 1, nbuf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ph == PHASES - 1) {   // slab kt + 1 has landed
                    if (pend_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PEND) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_LOOP) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                bf16x8_t af[4], bfr[NT];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8_t, av[i]);
#pragma unroll
                for (int j = 0; j < NT; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, bv[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[4 * ph + i][j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[4 * ph + i][j], 0, 0, 0);   // D = W X^T
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        if (group == 0) __builtin_amdgcn_s_barrier();   // re-align the groups
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy slabs past KT have landed (in the ring, not in the image)
        __builtin_amdgcn_s_barrier();

        // ---------------------------------------------------------------- epilogue: accumulators -> bf16 image
        unsigned char *img = smem;
        if (fused) {
            // conv followed by GDN1 / inverse GDN1: x -> image, norm = gamma |x| as a second GEMM (gamma fragment-major
            // from L2 into registers, one step ahead), y = x * (beta + norm) or x / (...) in place in the image
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int r = wm * C::WM + i * 16 + frow;
                    const int col = wn * C::WN + j * 16 + fq * 4;
                    uint2 h;
                    h.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                    h.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                    *reinterpret_cast<uint2 *>(img + Img::off(r, col >> 3) + (col & 7) * 2) = h;
                    acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            int ln = lane;
            asm volatile("" : "+v"(ln));   // (addresses rebuilt per tile, not hoisted out of the tile loop and spilled)
            const uint4 *gfrag = reinterpret_cast<const uint4 *>(p.ep_x) + (long long)(wn * NT) * (BN / 32) * 64 + ln;
            constexpr int NS = BN / 32;
            uint4 gbuf[2][NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) gbuf[0][j] = gfrag[(j * NS + 0) * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // the x image is complete
            const unsigned char *xrow = img + (wm * C::WM + frow) * 512;
#pragma unroll
            for (int ks2 = 0; ks2 < NS; ++ks2) {
                if (ks2 + 1 < NS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) gbuf[(ks2 + 1) & 1][j] = gfrag[(j * NS + ks2 + 1) * 64];
                }
                const int xc = ((4 * ks2 + fq) ^ frow) << 4;   // row & 15 == frow for every fragment row of this lane
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    uint4 v = *reinterpret_cast<const uint4 *>(xrow + i * 16 * 512 + xc);
                    v.x &= 0x7FFF7FFFu; v.y &= 0x7FFF7FFFu; v.z &= 0x7FFF7FFFu; v.w &= 0x7FFF7FFFu;
                    const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, gbuf[ks2 & 1][j]), af,
                                                                            acc[i][j], 0, 0, 0);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave is done with its x-image fragments
            const bool inverse = p.epi == SC2_EPI_FUSED_IGDN;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 bj = *reinterpret_cast<const float4 *>(p.ep_beta + wn * C::WN + j * 16 + fq * 4);
                const float b[4] = {bj.x, bj.y, bj.z, bj.w};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = wm * C::WM + i * 16 + frow;
                    const int col = wn * C::WN + j * 16 + fq * 4;
                    unsigned char *slot = img + Img::off(row, col >> 3) + (col & 7) * 2;
                    const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                    const float xv[4] = {__builtin_bit_cast(float, xr.x << 16), __builtin_bit_cast(float, xr.x & 0xFFFF0000u),
                                         __builtin_bit_cast(float, xr.y << 16), __builtin_bit_cast(float, xr.y & 0xFFFF0000u)};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float norm = b[e] + acc[i][j][e];
                        v[e] = inverse ? xv[e] * norm : xv[e] * (1.0f / norm);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(slot) = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = wm * C::WM + i * 16 + frow;
                    const int col = wn * C::WN + j * 16 + fq * 4;
                    uint2 o;
                    o.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                    o.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                    *reinterpret_cast<uint2 *>(img + Img::off(row, col >> 3) + (col & 7) * 2) = o;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the image is complete
        // ---- image -> registers (whole 16-byte channel runs, as they will be stored)
        {
            int tq = tid;
            asm volatile("" : "+v"(tq));
            SC2_PEND_LIST(SC2_PEND_LOAD)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has read its part: the LDS is free for the next tile's ring
        has_pend = true;
        pend_m0 = m0;
    }
    if (has_pend) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        SC2_PEND_LIST(SC2_PEND_STORE)
    }
#undef SC2_PEND_LIST
#undef SC2_PEND_DECL
#undef SC2_PEND_STORE
#undef SC2_PEND_LOAD
}

template <class C, int MODE>
int launch8p(const ConvArgs &a, hipStream_t s) {
    constexpr bool PIPE = MODE == 1;
    ConvArgs p = a;
    p.KT = (a.KH * a.KW * a.Cin + C::BK - 1) / C::BK;
    if (PIPE && (p.KT & 1)) {
        sc2_set_error("conv2d: the pipelined persistent kernel needs an even number of k-slabs");
        return SC2_ERR_UNSUPPORTED;
    }
    p.n_ntiles = 1;
    const int n_tiles = (a.M + C::BM - 1) / C::BM;
    // 128 KB: the ring, reused as the output image.  (The per-tile kernel asks for 160 KB -- its f32-output staging -- which
    // no other workgroup fits beside; with 128 KB an encode wave of the range coder (20 KB) can share the CU.)
    constexpr int LDS_P = C::STAGES * C::STAGE_BYTES;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    const int n_cus = sc2_device_cus();
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm8p_kernel<C, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
        attr_set = true;
    }
    // tiles per workgroup: SC2_CONV_CHUNK (0 / unset: the default below; large: one static share per CU)
    const int ce = sc2_pol().conv_chunk;
    const int per_xcd = (n_tiles + 7) / 8, cus_x = n_cus / 8 > 0 ? n_cus / 8 : 1;
    int chunk = ce > 0 ? ce : 2;   // measured inside the pipelined bench (tools/attic/chunk_ab.sh): 2 - 3 best, 1 static share per CU worst
    const int full = (per_xcd + cus_x - 1) / cus_x;          // the share of one CU
    if (chunk > full) chunk = full;
    const int grid = 8 * ((per_xcd + chunk - 1) / chunk);
    hipLaunchKernelGGL((conv_igemm8p_kernel<C, MODE>), dim3((unsigned)grid), dim3(512), LDS_P, s, p, n_tiles, chunk);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

template int launch8p<B_dec2, 0>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec4, 0>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec2, 1>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec4, 1>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec2, 2>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec4, 2>(const ConvArgs &, hipStream_t);
#ifdef SC2_EXPERIMENTS   // MODE 3 = timing experiment with garbage results: not in the shipped library
template int launch8p<B_dec2, 3>(const ConvArgs &, hipStream_t);
template int launch8p<B_dec4, 3>(const ConvArgs &, hipStream_t);
#endif

}  // namespace sc2conv
