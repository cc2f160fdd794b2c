// Streaming 1x1 convolution with a short K and a wide N for gfx950: y = act(x W^T + bias [+ residual]), bf16 NHWC.
// The caller-side layers of the bottleneck path that are pure HBM streams: the third conv of a torchvision Bottleneck
// block (128 -> 512 at 28^2, 256 -> 1024 at 14^2, with the residual add + ReLU) and the stride-2 1x1 downsample
// (sc2bench/models/backbone.py:235-254 runs them as layer2 / layer3 of ResNet-50).
//
// Why a dedicated kernel: on the generic tile kernel these launches are four to eight k-slabs of MFMA work wrapped in
// "load operand -> compute -> store -> wait for the acks"; a CU then has 2-3 workgroups x 32 KB in flight and the
// launch runs at 2.5 TB/s.  Here one persistent 512-thread workgroup per CU streams UNITS of 128 pixels x 256 output
// channels (64 KB out, 64 KB residual in): the next unit's loads (its A tile, its weight fragments, its residual
// tile) are issued as soon as the accumulators of the current unit are dead and BEFORE the current unit's output
// stores (vmcnt retires in issue order), so loads, MFMAs and stores of neighbouring units overlap.
//   A tile  [128 px][K] bf16 in LDS (<= 64 KB, 16-byte chunks XOR-swizzled by row), fragments by ds_read_b128
//   W       fragment-major ([16-channel tile][32-deep step][lane][8 k], 1 KB contiguous per MFMA operand), straight
//           from L2 into registers: the whole chunk (K/32 x 2 fragments per wave) before the K loop
//   image   [128 px][256 ch] bf16 in LDS (64 KB): the residual tile is parked there with coalesced 16-byte accesses,
//           every lane updates its own 8-byte slots in place in f32, the image is streamed out in 16-byte stores
// Units are claimed with one atomic each, one unit ahead.
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "sc2_common.h"

#ifndef SC2_STREAM_EARLY
#define SC2_STREAM_EARLY 1
#endif
#ifndef SC2_NT_STREAM
#define SC2_NT_STREAM 0   // non-temporal output stores: measured SLOWER here (the consumer launch finds part of this map in L2 / the memory-side cache: head + 2.5 %, dec.conv2 + 2 %); 1: A/B
#endif

namespace {

struct StreamArgs {
    const uint16_t *__restrict__ x;       // bf16 NHWC [N,H,W,K]
    const uint16_t *__restrict__ w;       // bf16 fragment-major [Cout/16][K/32][64][8]
    const float *__restrict__ bias;       // f32 [Cout]
    const uint16_t *__restrict__ res;     // bf16 NHWC [N,OH,OW,Cout] or null
    uint16_t *__restrict__ y;             // bf16 NHWC [N,OH,OW,Cout]
    const uint16_t *__restrict__ mask;    // bf16 like y or null (MASK instantiations): y = mask > 0 ? value : 0
    int H, W, OH, OW, OHW, M, Cout, stride, relu;
    int n_chunks, n_units;                // 256-channel chunks per pixel tile; units = pixel tiles x chunks
    unsigned *unit_ctr;                   // claims so far (claim c = unit c + 2 * gridDim.x); zero between launches
};

// Unit shapes (pixels x channels): 128 x 256 for K in {128, 256} when Cout % 256 == 0, 128 x 128 for the narrow layers,
// 64 x 128 for K = 512 (its weights - 16 channels x 512 per wave - still fit the registers: 64 VGPRs).

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(f32x2_t v) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// The tail of a unit (next unit's loads, this unit's stores) is straight-line code: addresses are clamped instead of
// guarded, so the compiler's vmcnt bookkeeping stays exact and its wait for the loads is vmcnt(<stores issued after
// them>) -- never a wait for a store acknowledgement.
// MASK (round 5): the ReLU gradient behind a data gradient rides in the store pass -- y = mask > 0 ? acc + bias [+ residual] : 0 with
// `mask` = the saved output of that ReLU (frozen.py: conv1's data gradient of a Bottleneck block + the skip path's gradient, masked
// by the previous block's output: the separate relu_bwd pass over the 4C-wide tensor disappears).  An instantiation of its own.
template <int K, int BM, int BNC, bool RES, bool MASK = false>
__global__ __launch_bounds__(512, 2) void conv1x1_stream_kernel(const StreamArgs p) {
    constexpr int MT = BM / 16, NT = BNC / 16 / 8;   // per wave: all pixel tiles x its 16 NT channels
    constexpr int CPI = BNC / 8;               // 16-byte chunks per image row
    constexpr int KS = K / 32;                 // k-steps
    constexpr int A_BYTES = BM * K * 2;        // A tile
    constexpr int CPR = K / 8;                 // 16-byte chunks per A row
    constexpr int A_Q = BM * CPR / 512;        // A chunks per thread
    constexpr int IMG_Q = BM * CPI / 512;      // image chunks per thread
    constexpr int ASW = CPR >= 16 ? 15 : CPR - 1;   // chunk XOR mask of the A tile (K = 64: eight chunks per row)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int next_slot;
    unsigned char *At = smem;
    unsigned char *img = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cout = p.Cout;

    // register sets of the NEXT unit, filled while the current one is being stored
    u32x4_t a_next[A_Q], r_next[RES ? IMG_Q : 1];   // native vectors: a struct copy becomes a memcpy SROA will not split
    uint4 w_regs[KS][NT];
    float4 b_next[NT];

    // PART 0: everything (prologue); 1: all but the bias (issued in front of the epilogue, see the unit loop); 2: the bias, which
    // the epilogue of the CURRENT unit still reads from b_next
    auto load_unit = [&](int unit, int tid, auto part_c) {
        constexpr int PART = decltype(part_c)::value;
        const bool live = unit < p.n_units;     // a dead unit loads unit 0's operands and never uses them
        const int tile = live ? unit / p.n_chunks : 0;
        const int chunk = live ? unit - tile * p.n_chunks : 0;
        const int m0 = tile * BM;
        if (PART != 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b_next[j] = *reinterpret_cast<const float4 *>(p.bias + chunk * BNC + wn * (NT * 16) + j * 16 + ((tid & 63) >> 4) * 4);
        }
        if (PART == 2) return;
        // weights of this wave's 32 channels of the chunk: all k-steps
        const uint4 *wf = reinterpret_cast<const uint4 *>(p.w) + ((long long)(chunk * (BNC / 16) + wn * NT) * KS) * 64 + (tid & 63);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w_regs[ks][j] = wf[(j * KS + ks) * 64];
        // residual tile (rows past M: clamped, their results are never stored)
        if (RES) {
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q / CPI, c = q % CPI;
                const int m = min(m0 + row, p.M - 1);
                r_next[k] = *reinterpret_cast<const u32x4_t *>(p.res + (long long)m * Cout + chunk * BNC + c * 8);
            }
        }
        // A tile: chunk q = tid + 512 k  ->  (row, 16-byte chunk); the youngest loads: waiting for them covers the rest
        // (ONE test of the stride per unit, not one per chunk: wave-uniform runtime conditions inside unrolled loops become
        //  scalar branches, and this kernel's unit loop held 62 of them for 128 MFMAs)
        if (p.stride != 1) {
#pragma unroll
            for (int k = 0; k < A_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q / CPR, c = q - row * CPR;
                const int m = min(m0 + row, p.M - 1);
                const int im = m / p.OHW, rem = m - im * p.OHW;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const long long pix = ((long long)im * p.H + oh * p.stride) * p.W + ow * p.stride;
                a_next[k] = *reinterpret_cast<const u32x4_t *>(p.x + pix * K + c * 8);
            }
        } else {
#pragma unroll
            for (int k = 0; k < A_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q / CPR, c = q - row * CPR;
                const long long pix = min(m0 + row, p.M - 1);
                a_next[k] = *reinterpret_cast<const u32x4_t *>(p.x + pix * K + c * 8);
            }
        }
    };
    auto store_a = [&](int tid) {
#pragma unroll
        for (int k = 0; k < A_Q; ++k) {
            const int q = tid + 512 * k;
            const int row = q / CPR, c = q - row * CPR;
            *reinterpret_cast<u32x4_t *>(At + row * (K * 2) + ((c ^ (row & ASW)) << 4)) = a_next[k];
        }
    };

    int unit = blockIdx.x;
    int next_unit = unit + gridDim.x;
    load_unit(unit, tid, std::integral_constant<int, 0>{});
    store_a(tid);
    __syncthreads();

    while (unit < p.n_units) {
        const int tile = unit / p.n_chunks;
        const int chunk = unit - tile * p.n_chunks;
        const int m0 = tile * BM;

        // ---- K loop: A fragments from LDS, weights from the registers loaded one unit ago
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int a_lane = frow * (K * 2) + (((ks * 4 + fq) ^ (frow & ASW)) << 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(At + a_lane + i * 16 * K * 2));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w_regs[ks][j]), af,
                                                                        acc[i][j], 0, 0, 0);
            }
        }
        // ---- residual of this unit (fetched one unit ago) into the image
        if (RES) {
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q / CPI, c = q % CPI;
                *reinterpret_cast<u32x4_t *>(img + row * (BNC * 2) + ((c ^ (row & 15)) << 4)) = r_next[k];
            }
        }
        // The NEXT unit's A tile, residual tile and weights are fetched HERE, in front of the epilogue, where they used to be
        // issued directly in front of this unit's output stores (round 4: loads queued in front of a store burst delay it;
        // their registers are free: a_next went to LDS at the end of the previous unit, r_next just above).
        // (the K = 512 x 256-channel residual instantiation keeps the late fetch: with it here it spilled)
        constexpr bool EARLY = SC2_STREAM_EARLY && !(K == 512 && BNC == 256 && RES);
        if (EARLY) load_unit(next_unit, tid, std::integral_constant<int, 1>{});
        __syncthreads();   // the A tile has been consumed by every wave; the residual image is complete
        // ---- y = act(acc + bias [+ residual]) in place in the image; explicit (e0,e1)/(e2,e3) pairs: packed ops
        // (the ReLU flag is tested ONCE per unit: as `if (p.relu)` inside the loops it was two scalar branches per accumulator tile)
        auto finish = [&](auto relu_c) {
            constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wn * (NT * 16) + j * 16 + fq * 4;            // channel inside the chunk
                const float4 b4 = b_next[j];
                const f32x2_t b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = i * 16 + frow;
                    unsigned char *slot = img + row * (BNC * 2) + (((col >> 3) ^ frow) << 4) + (col & 7) * 2;
                    f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]} + b01;
                    f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]} + b23;
                    if (RES) {
                        const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                        v01 += f32x2_t{__builtin_bit_cast(float, xr.x << 16), __builtin_bit_cast(float, xr.x & 0xFFFF0000u)};
                        v23 += f32x2_t{__builtin_bit_cast(float, xr.y << 16), __builtin_bit_cast(float, xr.y & 0xFFFF0000u)};
                    }
                    if (RELU) {
                        v01 = f32x2_t{fmaxf(v01[0], 0.f), fmaxf(v01[1], 0.f)};
                        v23 = f32x2_t{fmaxf(v23[0], 0.f), fmaxf(v23[1], 0.f)};
                    }
                    uint2 o;
                    o.x = pack2(v01);
                    o.y = pack2(v23);
                    *reinterpret_cast<uint2 *>(slot) = o;
                }
            }
        };
        if (p.relu) finish(std::true_type{});
        else finish(std::false_type{});
        __syncthreads();
        // ---- the accumulators are dead: claim a unit and fetch the next unit's operands, THEN stream this unit out
        int tq = tid;   // opaque: per-thread offsets are recomputed here, not carried (spilled) across the unit
        asm volatile("" : "+v"(tq));
        unsigned claimed = 0;
        if (tid == 0) {   // raw instruction: the compiler's atomicAdd waits for the result (vmcnt(0)) on the spot
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(p.unit_ctr), "v"(one) : "memory");
        }
        if (EARLY) load_unit(next_unit, tq, std::integral_constant<int, 2>{});
        else load_unit(next_unit, tq, std::integral_constant<int, 0>{});
        {
            uint4 *yo = reinterpret_cast<uint4 *>(p.y + (long long)chunk * BNC);
            [[maybe_unused]] uint4 mq[MASK ? IMG_Q : 1];
            if constexpr (MASK) {
                const uint4 *mo = reinterpret_cast<const uint4 *>(p.mask + (long long)chunk * BNC);
#pragma unroll
                for (int k = 0; k < IMG_Q; ++k) {
                    const int q = tq + 512 * k;
                    int row = q / CPI;
                    const int c = q % CPI;
                    row = m0 + row < p.M ? row : 0;
                    mq[k] = mo[(long long)(m0 + row) * (Cout / 8) + c];
                }
            }
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tq + 512 * k;
                int row = q / CPI;
                const int c = q % CPI;
                row = m0 + row < p.M ? row : 0;      // past the end: row 0 of the tile again (same data, same address)
                uint4 ov = *reinterpret_cast<const uint4 *>(img + row * (BNC * 2) + ((c ^ (row & 15)) << 4));
                if constexpr (MASK) {
                    const uint32_t mw[4] = {mq[k].x, mq[k].y, mq[k].z, mq[k].w};
                    uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ow[e] = (__builtin_bit_cast(float, mw[e] << 16) > 0.f ? ow[e] & 0xFFFFu : 0u) |
                                (__builtin_bit_cast(float, mw[e] & 0xFFFF0000u) > 0.f ? ow[e] & 0xFFFF0000u : 0u);
                    ov = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
                if (SC2_NT_STREAM) sc2_store16_nt(yo + (long long)(m0 + row) * (Cout / 8) + c, ov);
                else yo[(long long)(m0 + row) * (Cout / 8) + c] = ov;
            }
        }
        store_a(tq);       // the A tile region was last read before the first barrier of this unit
        if (tid == 0) {    // the claim is older than the loads store_a() has just waited for
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(claimed) : "n"(IMG_Q) : "memory");
            next_slot = (int)(claimed + 2 * gridDim.x);
            if (claimed == (unsigned)(p.n_units - 1)) *p.unit_ctr = 0u;   // the launch's last claim re-arms the counter
        }
        __syncthreads();   // next A tile visible; image free
        unit = next_unit;
        next_unit = __builtin_amdgcn_readfirstlane(next_slot);
    }
}

constexpr int kRing = 256;
sc2_counter_ring g_ring;
std::atomic<unsigned> g_seq{0};

template <int K, int BM, int BNC, bool RES, bool MASK = false>
int launch_stream(const StreamArgs &a0, hipStream_t s) {
    constexpr int lds = BM * K * 2 + BM * BNC * 2;
    StreamArgs a = a0;
    a.n_chunks = a.Cout / BNC;
    const long long units = (((long long)a.M + BM - 1) / BM) * a.n_chunks;
    if (units >= 0x7FFFFFFFLL - 1024) {
        sc2_set_error("conv1x1_stream: too many units");
        return SC2_ERR_UNSUPPORTED;
    }
    a.n_units = (int)units;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1x1_stream_kernel<K, BM, BNC, RES, MASK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int g_cus = sc2_device_cus();
    unsigned *slot = g_ring.launch_slot(s, kRing, 1, g_seq);
    if (!slot) return SC2_ERR_INTERNAL;
    static sc2_per_device_int per_cu_dev;   // resident workgroups per CU of this instantiation (LDS and registers decide), per device
    int per_cu = per_cu_dev.here().load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&conv1x1_stream_kernel<K, BM, BNC, RES, MASK>), 512,
                                                         lds) != hipSuccess || n < 1)
            n = 1;
        per_cu = n > 2 ? 2 : n;
        per_cu_dev.here().store(per_cu, std::memory_order_relaxed);
    }
    const int slots = g_cus * per_cu;
    const int grid = a.n_units < slots ? a.n_units : slots;
    StreamArgs b = a;
    b.unit_ctr = slot;
    hipLaunchKernelGGL((conv1x1_stream_kernel<K, BM, BNC, RES, MASK>), dim3(grid), dim3(512), lds, s, b);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv1x1_stream_supported(int Cin, int Cout, int stride) {
    return (Cin == 64 || Cin == 128 || Cin == 256 || Cin == 512) && Cout >= 128 && Cout % 128 == 0 && (stride == 1 || stride == 2) ? 1 : 0;
}

extern "C" int sc2_conv1x1_stream_mask_supported(int Cin, int Cout, int stride) {
    return (Cin == 128 || Cin == 256) && Cout % 256 == 0 && Cout >= 256 && stride == 1 ? 1 : 0;
}

extern "C" int sc2_conv1x1_stream_fwd(const void *x, const void *w_frag, const float *bias, const void *residual, const void *mask,
                                      void *y, int N, int H, int W, int Cin, int Cout, int stride, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv1x1_stream: null argument");
    SC2_REQUIRE(!mask || (!relu && sc2_conv1x1_stream_mask_supported(Cin, Cout, stride)), SC2_ERR_UNSUPPORTED,
                "conv1x1_stream: the mask form needs Cin 128 / 256, Cout %% 256 == 0, stride 1, no relu (got %d -> %d, stride %d)", Cin, Cout,
                stride);
    SC2_REQUIRE(N > 0 && H > 0 && W > 0, SC2_ERR_INVALID_ARG, "conv1x1_stream: non-positive dimension");
    SC2_REQUIRE(sc2_conv1x1_stream_supported(Cin, Cout, stride), SC2_ERR_UNSUPPORTED,
                "conv1x1_stream: needs Cin in {64, 128, 256, 512}, Cout %% 128 == 0, stride 1 or 2 (got %d -> %d, stride %d)", Cin,
                Cout, stride);
    StreamArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.res = static_cast<const uint16_t *>(residual);
    a.mask = static_cast<const uint16_t *>(mask);
    a.y = static_cast<uint16_t *>(y);
    a.H = H; a.W = W;
    a.OH = (H - 1) / stride + 1;
    a.OW = (W - 1) / stride + 1;
    a.OHW = a.OH * a.OW;
    const long long M = (long long)N * a.OHW;
    SC2_REQUIRE(M < 0x7FFFFFFFLL - 256, SC2_ERR_UNSUPPORTED, "conv1x1_stream: N*OH*OW = %lld exceeds 2^31", M);
    a.M = (int)M; a.Cout = Cout; a.stride = stride; a.relu = relu ? 1 : 0;
    a.n_chunks = 0; a.n_units = 0;   // set per unit shape by the launcher
    a.unit_ctr = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // unit shape: K = 512 -> 64 x 256 (or 64 x 128); otherwise 128 x 256 if it divides Cout, else 128 x 128
    if (a.mask) {   // (data gradient of a block's conv1: 128 -> 512 / 256 -> 1024, with or without the skip path's gradient)
        if (Cin == 128) return a.res ? launch_stream<128, 128, 256, true, true>(a, s) : launch_stream<128, 128, 256, false, true>(a, s);
        return a.res ? launch_stream<256, 128, 256, true, true>(a, s) : launch_stream<256, 128, 256, false, true>(a, s);
    }
#define SC2_STREAM_GO(KK, BMM, BNN) return a.res ? launch_stream<KK, BMM, BNN, true>(a, s) : launch_stream<KK, BMM, BNN, false>(a, s)
    if (Cin == 512 && Cout % 256 == 0) SC2_STREAM_GO(512, 64, 256);   // (half the A re-reads of the 128-wide unit)
    if (Cin == 512) SC2_STREAM_GO(512, 64, 128);
    // K = 64: conv3 / downsample of layer1 (64 -> 256 at 56 x 56; only a frozen TEACHER runs them: the student's layer1 is the bottleneck)
    if (Cin == 64 && Cout % 256 == 0) SC2_STREAM_GO(64, 128, 256);
    if (Cin == 64) SC2_STREAM_GO(64, 128, 128);
    if (Cout % 256 == 0) {   // (64-pixel units, two workgroups per CU: measured the same for K = 128, 5 - 15 % slower for K = 256)
        if (Cin == 128) SC2_STREAM_GO(128, 128, 256);
        SC2_STREAM_GO(256, 128, 256);
    }
    if (Cin == 128) SC2_STREAM_GO(128, 128, 128);
    SC2_STREAM_GO(256, 128, 128);
#undef SC2_STREAM_GO
}
