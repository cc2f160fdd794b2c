// 1x1 convolution with a LONG K and the weights resident in registers (gfx950): the conv1 layers of the ResNet tail's
// layer3 / layer4.0 (1024 -> 256 / 512, torchvision Bottleneck.conv1 + bn1 + ReLU in eval mode, backbone.py:235-254):
//     y[m, n] = act( sum_k x[m, k] w[n, k] + bias[n] ),  K = 1024, bf16 NHWC in and out.
//
// M is small there (50 176 pixels at bs 256) and the weights are 0.5 - 1 MB: a tile kernel re-reads them per 128-pixel
// tile (205 MB through L2 per launch, 355 TFLOP/s).  Here a workgroup is bound to ONE 128-channel chunk of the output
// for its whole life and keeps that chunk's weights in registers:
//   * 4 waves, one per SIMD, 512 registers each; wave (kh, nh) holds K half kh x 64 channels nh of the chunk:
//     16 k-steps x 4 channel tiles = 64 fragments = 256 VGPRs, loaded once;
//   * a unit = 32 pixels: its A tile (64 KB) arrives by buffer-addressed LDS-DMA into a double buffer (next unit's
//     pieces are issued behind the first k-steps of the current one), each wave multiplies its K half for both 16-pixel
//     tiles (128 MFMAs) reading the A fragments of k-step s + 1 between the MFMAs of k-step s;
//   * the two K halves are exchanged through 16 KB of LDS: wave (kh, nh) sends the pixel tile it does not own and adds
//     its partner's partial for pixel tile kh; bias + ReLU, bf16, 8 KB output image, streamed out;
//   * units are claimed dynamically, two ahead, from one counter per (XCD, chunk): an XCD owns the pixel tiles
//     t = xcd (mod 8) for every chunk, so the chunks' workgroups read a tile's A rows through the same L2 at about the
//     same time (A from HBM once, not once per chunk); a workgroup whose CU is held by another kernel takes fewer.
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "sc2_common.h"

#ifndef SC2_NT_KRES
#define SC2_NT_KRES 0   // non-temporal output stores: measured SLOWER here (the consumer launch finds part of this map in L2 / the memory-side cache: head + 2.5 %, dec.conv2 + 2 %); 1: A/B
#endif

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

struct KresArgs {
    const uint16_t *__restrict__ x;      // bf16 NHWC [N, H, W, K]
    const uint16_t *__restrict__ w;      // bf16 fragment-major [Cout/16][K/32][64][8]
    const float *__restrict__ bias;      // f32 [Cout]
    uint16_t *__restrict__ y;            // bf16 [M, Cout],  M = N * OH * OW
    int M, Cout, relu, n_chunks, n_tiles;
    int H, W, OW, OHW, stride;           // stride 2: output pixel (im, oh, ow) reads input pixel (im, 2 oh, 2 ow)
    unsigned in_bytes;                   // size of x
    unsigned *unit_ctr;                  // counters [8 XCDs][16 chunks]
};

// Shapes: K = 1024 -> units of 32 pixels x 128 channels; K = 2048 -> 16 pixels x 64 channels.  Either way a wave holds
// K / 2 x BNC / 2 weights = 256 VGPRs and an A tile is 64 KB.
template <int K_, int BM_, int BNC_>
struct KresShape {
    static constexpr int K = K_, BM = BM_, BNC = BNC_;
    static constexpr int KS = K / 32, KSH = KS / 2;          // k-steps, per K half
    static constexpr int MT = BM / 16, NTW = BNC / 32;       // pixel tiles, channel tiles per wave (its half of the chunk)
    static constexpr int NB = MT * NTW, NOWN = NB / 2;       // accumulator blocks per wave, owned after the exchange
    static constexpr int A_BYTES = BM * K * 2;               // 65 536
    static constexpr int PPR = K * 2 / 1024;                 // 1 KB pieces per A row
    static constexpr int PIECES = A_BYTES / 1024, PIECES_PER_WAVE = PIECES / 4;   // 64, 16
    static constexpr int XCH_OFF = 2 * A_BYTES;              // exchange area [4 waves][NOWN blocks][64 lanes][16 B]
    static constexpr int OIMG_OFF = XCH_OFF;                 // output image [BM][BNC * 2 B], over the dead exchange area
    static constexpr int LDS_BYTES = XCH_OFF + 16 * 1024;
    static constexpr int OUT_CHUNKS = BM * BNC * 2 / 16;     // 16-byte chunks of a unit's output
    static constexpr int STORES = OUT_CHUNKS >= 256 ? OUT_CHUNKS / 256 : 1;   // store instructions per thread and unit
    static_assert(A_BYTES == 65536 && (K / 64) * (BNC / 32) * 4 == 256, "64 KB A tile, 256 weight registers per wave");
    static_assert(4 * NOWN * 1024 <= 16 * 1024 && BM * BNC * 2 <= 16 * 1024 && STORES >= 1, "exchange / image area");
    static_assert(LDS_BYTES + 64 <= 160 * 1024, "one workgroup per CU");
};

template <class S>
__global__ __launch_bounds__(256, 1) void conv1x1_kres_kernel(const KresArgs p) {
    constexpr int K = S::K, BM = S::BM, BNC = S::BNC, KS = S::KS, KSH = S::KSH, MT = S::MT, NTW = S::NTW, NOWN = S::NOWN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int next_slot;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, nh = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // Workgroup b runs on XCD b & 7 (round-robin dispatch).  Each XCD takes the pixel tiles t = xcd (mod 8) for EVERY
    // chunk, and its workgroups are dealt to the chunks in turn: the workgroups of the different chunks walk the same
    // tile sequence at the same pace inside one L2, so a tile's A rows come from HBM once, not once per chunk.
    const int xcd = blockIdx.x & 7;
    const int j_x = blockIdx.x >> 3;                          // index of this workgroup on its XCD
    const int chunk = j_x % p.n_chunks;                       // its BNC output channels, for its whole life
    const int wg_l = j_x / p.n_chunks;                        // index among the XCD's workgroups of that chunk
    const int wgs_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int wgs_c = (wgs_x - chunk + p.n_chunks - 1) / p.n_chunks;
    const int n_local = (p.n_tiles - xcd + 7) >> 3;           // tiles of this XCD: local tile u = global tile xcd + 8 u
    unsigned *const my_ctr = p.unit_ctr + xcd * 16 + chunk;

    // ---- resident weights: k-steps [KSH kh, KSH kh + KSH) x channel tiles [NTW nh, NTW nh + NTW) of the chunk
    uint4 wreg[KSH][NTW];
    {
        const uint4 *wf = reinterpret_cast<const uint4 *>(p.w) + lane;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int s = 0; s < KSH; ++s)
                wreg[s][j] = wf[((long long)(chunk * (BNC / 16) + nh * NTW + j) * KS + kh * KSH + s) * 64];
    }

    // ---- A tile fill: piece pc = wave + 4 k = (row pc / PPR, part pc % PPR) = physical chunks [64 part, 64 part + 64) of
    //      the row; physical chunk c holds logical chunk c ^ (row & 15) (conflict-free fragment reads)
    const buf_rsrc_t rs = make_rsrc(p.x, p.in_bytes);
    // input pixel of output pixel m (stride 2: every other pixel of every other row), as a byte offset; lane r < BM of
    // every wave computes row r of a tile, the pieces pick theirs with v_readlane
    auto row_offsets = [&](int tile) -> uint32_t {
        const int m = (xcd + 8 * tile) * BM + (lane & (BM - 1));
        const bool ok = tile < n_local && m < p.M;
        long long pix = m;
        if (p.stride == 2) {
            const int mm = ok ? m : 0;
            const int im = mm / p.OHW, rem = mm - im * p.OHW;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            pix = ((long long)im * p.H + 2 * oh) * p.W + 2 * ow;
        }
        return ok ? (uint32_t)(pix * (K * 2)) : 0xFFFFFFFFu;
    };
    auto issue_piece = [&](uint32_t rowoff_v, int buf, int k) {
        const int pc = wave + 4 * k;
        const int row = pc / S::PPR, part = pc % S::PPR;
        const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane((int)rowoff_v, row);   // scalar
        const bool ok = ro != 0xFFFFFFFFu;
        const uint32_t vo = ok ? (uint32_t)(((64 * part + lane) ^ (row & 15)) * 16) : 0x80000000u;
        buf_load_lds16(rs, (lds_ptr_t)(smem + buf * S::A_BYTES + pc * 1024), vo, ok ? ro : 0u);
    };

    int unit = wg_l;
    int next_unit = unit + wgs_c;
    {
        const uint32_t ro = row_offsets(unit);
#pragma unroll
        for (int k = 0; k < S::PIECES_PER_WAVE; ++k) issue_piece(ro, 0, k);
    }
    int g = 0;   // units done by this workgroup: unit g's A tile lives in buffer g & 1

    while (unit < n_local) {
        const int m0 = (xcd + 8 * unit) * BM;
        // this unit's A tile has landed (this wave's pieces), then everybody's.  (Behind the first unit only the output
        // stores of the previous unit were issued after these pieces.)
        if (g != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned claimed = 0;
        if (tid == 0) {   // raw instruction: the compiler's atomicAdd waits for the result on the spot (tools/audit_asm_atomic.py)
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(my_ctr), "v"(one) : "memory");
        }
        const uint32_t ro_next = row_offsets(next_unit);
        const uint32_t ab = lds_base + (uint32_t)((g & 1) * S::A_BYTES);
        // fragment of pixel tile i at k-step ks: row i * 16 + frow, logical chunk ks * 4 + fq
        const uint32_t a_lane = ab + (uint32_t)(frow * (K * 2));
        f32x4_t acc[MT][NTW];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint4 av[2][MT];
        {
            int fqo = fq;
            asm volatile("" : "+v"(fqo));
            const int ks = kh * KSH;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                av[0][i] = lds_read16(a_lane + (uint32_t)(i * 16 * K * 2) + (uint32_t)((((ks * 4 + fqo) ^ frow)) << 4));
        }
        constexpr int PIECE_STEPS = 8;                                     // k-steps that carry the next unit's pieces
        constexpr int PER_STEP = S::PIECES_PER_WAVE / PIECE_STEPS;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < KSH) {
                int fqo = fq;
                asm volatile("" : "+v"(fqo));   // (addresses rebuilt per step, not hoisted and spilled)
                const int ks = kh * KSH + s + 1;
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    av[(s + 1) & 1][i] = lds_read16(a_lane + (uint32_t)(i * 16 * K * 2) + (uint32_t)((((ks * 4 + fqo) ^ frow)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, av[s & 1][i]);
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[s][j]), af, acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next unit's A tile: two pieces behind each of the first eight k-steps
            if (s < PIECE_STEPS) {
#pragma unroll
                for (int q = 0; q < PER_STEP; ++q) issue_piece(ro_next, (g + 1) & 1, PER_STEP * s + q);
            }
        }
        // ---------------------------------------------------------------- exchange of the K halves: block b = i NTW + j;
        // wave (kh, nh) keeps blocks [NOWN kh, NOWN kh + NOWN) and sends the others to its partner (1 - kh, nh)
#pragma unroll
        for (int b = 0; b < NOWN; ++b) {
            const f32x4_t v = kh ? acc[b / NTW][b % NTW] : acc[(NOWN + b) / NTW][(NOWN + b) % NTW];   // the block I do not own
            *reinterpret_cast<f32x4_t *>(smem + S::XCH_OFF + ((wave * NOWN + b) * 64 + lane) * 16) = v;
        }
        __syncthreads();
        f32x4_t own[NOWN];
#pragma unroll
        for (int b = 0; b < NOWN; ++b) {
            const f32x4_t mine = kh ? acc[(NOWN + b) / NTW][(NOWN + b) % NTW] : acc[b / NTW][b % NTW];
            own[b] = mine + *reinterpret_cast<const f32x4_t *>(smem + S::XCH_OFF + (((wave ^ 2) * NOWN + b) * 64 + lane) * 16);
        }
        __syncthreads();   // the exchange area becomes the output image
        // ---------------------------------------------------------------- bias (+ ReLU), bf16, output image [BM][BNC * 2 B]
        {
            unsigned char *oimg = smem + S::OIMG_OFF;
            constexpr int CPI = BNC / 8;
            // (the ReLU flag is tested once per unit: inside the loop it was two scalar branches per owned tile, on a wave that has
            //  its SIMD to itself)
            auto finish = [&](auto relu_c) {
                constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
                for (int b = 0; b < NOWN; ++b) {
                    const int bb = kh * NOWN + b;
                    const int i = bb / NTW, j = bb % NTW;             // (scalar)
                    const int px = i * 16 + frow;
                    const int ch = nh * (BNC / 2) + j * 16 + fq * 4;  // channel inside the chunk
                    const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + chunk * BNC + ch);
                    f32x2_t v01 = f32x2_t{own[b][0], own[b][1]} + f32x2_t{bias4.x, bias4.y};
                    f32x2_t v23 = f32x2_t{own[b][2], own[b][3]} + f32x2_t{bias4.z, bias4.w};
                    if (RELU) {
                        v01 = f32x2_t{fmaxf(v01[0], 0.f), fmaxf(v01[1], 0.f)};
                        v23 = f32x2_t{fmaxf(v23[0], 0.f), fmaxf(v23[1], 0.f)};
                    }
                    uint2 o;
                    o.x = pack2(v01);
                    o.y = pack2(v23);
                    const int c = ch >> 3;                            // 16-byte chunk of the row; swizzled by the row
                    *reinterpret_cast<uint2 *>(oimg + px * (BNC * 2) + ((c ^ (px & (CPI - 1))) << 4) + (ch & 7) * 2) = o;
                }
            };
            if (p.relu) finish(std::true_type{});
            else finish(std::false_type{});
        }
        __syncthreads();
        {
            const unsigned char *oimg = smem + S::OIMG_OFF;
            uint4 *yo = reinterpret_cast<uint4 *>(p.y + (long long)chunk * BNC);
            constexpr int CPI = BNC / 8;                          // 16-byte chunks per image row
#pragma unroll
            for (int k = 0; k < S::STORES; ++k) {
                const int q = (tid + 256 * k) % S::OUT_CHUNKS;   // (a 2 KB image: both halves of the workgroup store it)
                int row = q / CPI;
                const int c = q % CPI;
                row = m0 + row < p.M ? row : 0;      // past the end: row 0 of the tile again (same data, same address)
                const uint4 ov = *reinterpret_cast<const uint4 *>(oimg + row * (BNC * 2) + ((c ^ (row & (CPI - 1))) << 4));
                if (SC2_NT_KRES) sc2_store16_nt(yo + (long long)(m0 + row) * (p.Cout / 8) + c, ov);
                else yo[(long long)(m0 + row) * (p.Cout / 8) + c] = ov;
            }
        }
        if (tid == 0) {   // the claim is older than the A pieces whose wait opens the next unit - and than these stores
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(claimed) : "n"(S::STORES) : "memory");
            next_slot = (int)(claimed + 2 * wgs_c);
            if (claimed == (unsigned)(n_local - 1)) *my_ctr = 0u;   // the last claim of this (XCD, chunk) re-arms its counter
        }
        __syncthreads();
        ++g;
        unit = next_unit;
        next_unit = __builtin_amdgcn_readfirstlane(next_slot);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile prefetched for a unit that does not exist
}

constexpr int kRingK = 1024;
sc2_counter_ring g_ring_k;
std::atomic<unsigned> g_seq_k{0};

template <class S>
int launch_kres(KresArgs a, hipStream_t s) {
    a.n_chunks = a.Cout / S::BNC;
    a.n_tiles = (a.M + S::BM - 1) / S::BM;
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1x1_kres_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  S::LDS_BYTES);
        attr_set = true;
    }
    const int g_cus_k = sc2_device_cus();
    unsigned *slot = g_ring_k.launch_slot(s, kRingK, 128, g_seq_k);   // [8 XCDs][16 chunks] counters per launch
    if (!slot) return SC2_ERR_INTERNAL;
    const long long units = (long long)a.n_tiles * a.n_chunks;
    int grid = units < g_cus_k ? (int)units : g_cus_k;          // one 4-wave workgroup per CU
    if (grid < 8 * a.n_chunks) grid = 8 * a.n_chunks;           // every (XCD, chunk) needs a workgroup
    a.unit_ctr = slot;
    hipLaunchKernelGGL(conv1x1_kres_kernel<S>, dim3(grid), dim3(256), S::LDS_BYTES, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv1x1_kres_supported(int Cin, int Cout, int stride) {
    if (stride != 1 && stride != 2) return 0;
    if (Cin == 1024) return Cout >= 128 && Cout % 128 == 0 && Cout <= 16 * 128 ? 1 : 0;
    if (Cin == 2048) return Cout >= 64 && Cout % 64 == 0 && Cout <= 16 * 64 ? 1 : 0;
    return 0;
}

extern "C" int sc2_conv1x1_kres_fwd(const void *x, const void *w_frag, const float *bias, void *y, int N, int H, int W, int Cin,
                                    int Cout, int stride, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv1x1_kres: null argument");
    SC2_REQUIRE(N > 0 && H > 0 && W > 0, SC2_ERR_INVALID_ARG, "conv1x1_kres: non-positive dimension");
    SC2_REQUIRE(sc2_conv1x1_kres_supported(Cin, Cout, stride), SC2_ERR_UNSUPPORTED,
                "conv1x1_kres: needs Cin 1024 (Cout %% 128 == 0, <= 2048) or 2048 (Cout %% 64 == 0, <= 1024), stride 1 or 2 "
                "(got %d -> %d, stride %d)", Cin, Cout, stride);
    const long long in_bytes = (long long)N * H * W * Cin * 2;
    SC2_REQUIRE(in_bytes < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "conv1x1_kres: input of %lld bytes exceeds 2 GB", in_bytes);
    KresArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.H = H; a.W = W; a.stride = stride;
    const int OH = (H - 1) / stride + 1;
    a.OW = (W - 1) / stride + 1;
    a.OHW = OH * a.OW;
    a.M = N * a.OHW; a.Cout = Cout; a.relu = relu ? 1 : 0;
    a.in_bytes = (unsigned)in_bytes;
    a.n_chunks = 0; a.n_tiles = 0; a.unit_ctr = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cin == 1024) return launch_kres<KresShape<1024, 32, 128>>(a, s);
    return launch_kres<KresShape<2048, 16, 64>>(a, s);
}
