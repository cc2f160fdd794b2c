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

#include "sc2_common.h"

namespace {

struct StreamArgs {
    const uint16_t *__restrict__ x;       // bf16 NHWC [N,H,W,K]
    const uint16_t *__restrict__ w;       // bf16 fragment-major [Cout/16][K/32][64][8]
    const float *__restrict__ bias;       // f32 [Cout]
    const uint16_t *__restrict__ res;     // bf16 NHWC [N,OH,OW,Cout] or null
    uint16_t *__restrict__ y;             // bf16 NHWC [N,OH,OW,Cout]
    int H, W, OH, OW, OHW, M, Cout, stride, relu;
    int n_chunks, n_units;                // 256-channel chunks per pixel tile; units = pixel tiles x chunks
    unsigned *unit_ctr;                   // next unclaimed unit; preset to 2 * gridDim.x on the stream
};

constexpr int BM = 128, BNC = 256, MT = 8, NT = 2;

template <int K>
__global__ __launch_bounds__(512, 2) void conv1x1_stream_kernel(const StreamArgs p) {
    constexpr int KS = K / 32;                 // k-steps
    constexpr int A_BYTES = BM * K * 2;        // A tile
    constexpr int CPR = K / 8;                 // 16-byte chunks per A row
    constexpr int A_Q = BM * CPR / 512;        // A chunks per thread
    constexpr int IMG_Q = BM * (BNC / 8) / 512;   // image chunks per thread (8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *At = smem;
    unsigned char *img = smem + A_BYTES;
    volatile int *next_slot = reinterpret_cast<volatile int *>(smem + A_BYTES + BM * BNC * 2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int Cout = p.Cout;

    // register sets of the NEXT unit, filled while the current one is being stored
    uint4 a_next[A_Q], r_next[IMG_Q], w_regs[KS][NT];

    auto load_unit = [&](int unit) {
        const bool live = unit < p.n_units;
        const int tile = live ? unit / p.n_chunks : 0;
        const int chunk = live ? unit - tile * p.n_chunks : 0;
        const int m0 = tile * BM;
        // weights of this wave's 32 channels of the chunk: all k-steps
        const uint4 *wf = reinterpret_cast<const uint4 *>(p.w) + ((long long)(chunk * (BNC / 16) + wn * NT) * KS) * 64 + lane;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w_regs[ks][j] = wf[(j * KS + ks) * 64];
        // A tile: chunk q = tid + 512 k  ->  (row, 16-byte chunk)
#pragma unroll
        for (int k = 0; k < A_Q; ++k) {
            const int q = tid + 512 * k;
            const int row = q / CPR, c = q - row * CPR;
            const int m = m0 + row;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (live && m < p.M) {
                long long pix = m;
                if (p.stride != 1) {
                    const int im = m / p.OHW, rem = m - im * p.OHW;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    pix = ((long long)im * p.H + oh * p.stride) * p.W + ow * p.stride;
                }
                v = *reinterpret_cast<const uint4 *>(p.x + pix * K + c * 8);
            }
            a_next[k] = v;
        }
        // residual tile
        if (p.res) {
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q >> 5, c = q & 31;
                const int m = m0 + row;
                r_next[k] = (live && m < p.M)
                                ? *reinterpret_cast<const uint4 *>(p.res + (long long)m * Cout + chunk * BNC + c * 8)
                                : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int k = 0; k < A_Q; ++k) {
            const int q = tid + 512 * k;
            const int row = q / CPR, c = q - row * CPR;
            *reinterpret_cast<uint4 *>(At + row * (K * 2) + ((c ^ (row & 15)) << 4)) = a_next[k];
        }
    };

    int unit = blockIdx.x;
    int next_unit = unit + gridDim.x;
    load_unit(unit);
    store_a();
    __syncthreads();

    while (unit < p.n_units) {
        const int tile = unit / p.n_chunks;
        const int chunk = unit - tile * p.n_chunks;
        const int m0 = tile * BM;
        if (tid == 0) *next_slot = (int)atomicAdd(p.unit_ctr, 1u);

        // ---- K loop: A fragments from LDS, weights from the registers loaded one unit ago
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int a_lane = frow * (K * 2) + (((ks * 4 + fq) ^ frow) << 4);
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
        if (p.res) {
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q >> 5, c = q & 31;
                *reinterpret_cast<uint4 *>(img + row * (BNC * 2) + ((c ^ (row & 15)) << 4)) = r_next[k];
            }
        }
        __syncthreads();   // the A tile has been consumed by every wave; the residual image is complete
        // ---- y = act(acc + bias [+ residual]) in place in the image
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wn * (NT * 16) + j * 16 + fq * 4;            // channel inside the chunk
            const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + chunk * BNC + col);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 16 + frow;
                unsigned char *slot = img + row * (BNC * 2) + (((col >> 3) ^ frow) << 4) + (col & 7) * 2;
                float v[4] = {acc[i][j][0] + b[0], acc[i][j][1] + b[1], acc[i][j][2] + b[2], acc[i][j][3] + b[3]};
                if (p.res) {
                    const uint2 xr = *reinterpret_cast<const uint2 *>(slot);
                    v[0] += __builtin_bit_cast(float, xr.x << 16);
                    v[1] += __builtin_bit_cast(float, xr.x & 0xFFFF0000u);
                    v[2] += __builtin_bit_cast(float, xr.y << 16);
                    v[3] += __builtin_bit_cast(float, xr.y & 0xFFFF0000u);
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(slot) = o;
            }
        }
        __syncthreads();
        // ---- the accumulators are dead: fetch the next unit's operands, THEN stream this unit out
        const int unit_after_next = __builtin_amdgcn_readfirstlane(*next_slot);
        load_unit(next_unit);
        {
            uint4 *yo = reinterpret_cast<uint4 *>(p.y);
#pragma unroll
            for (int k = 0; k < IMG_Q; ++k) {
                const int q = tid + 512 * k;
                const int row = q >> 5, c = q & 31;
                const int m = m0 + row;
                if (m < p.M)
                    yo[((long long)m * Cout + chunk * BNC) / 8 + c] =
                        *reinterpret_cast<const uint4 *>(img + row * (BNC * 2) + ((c ^ (row & 15)) << 4));
            }
        }
        store_a();         // the A tile region was last read before the first barrier of this unit
        __syncthreads();   // next A tile visible; image free
        unit = next_unit;
        next_unit = unit_after_next;
    }
}

int g_cus = 0;
constexpr int kMaxDev = 16, kRing = 256;
unsigned *g_ring[kMaxDev] = {};
std::atomic<unsigned> g_seq{0};

template <int K>
int launch_stream(const StreamArgs &a, hipStream_t s) {
    constexpr int lds = BM * K * 2 + BM * BNC * 2 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1x1_stream_kernel<K>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev) {
        sc2_set_error("conv1x1_stream: device ordinal %d out of range", dev);
        return SC2_ERR_UNSUPPORTED;
    }
    if (g_cus == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus = n;
    }
    if (!g_ring[dev]) {
        void *ptr = nullptr;
        if (hipMalloc(&ptr, kRing * sizeof(unsigned)) != hipSuccess) {
            sc2_set_error("conv1x1_stream: cannot allocate the unit counters");
            return SC2_ERR_INTERNAL;
        }
        g_ring[dev] = static_cast<unsigned *>(ptr);
    }
    const int grid = a.n_units < g_cus ? a.n_units : g_cus;
    StreamArgs b = a;
    b.unit_ctr = g_ring[dev] + (g_seq.fetch_add(1) % kRing);
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b.unit_ctr), 2 * grid, 1, s) != hipSuccess) {
        sc2_set_error("conv1x1_stream: cannot preset the unit counter");
        return SC2_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(conv1x1_stream_kernel<K>, dim3(grid), dim3(512), lds, s, b);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

extern "C" int sc2_conv1x1_stream_supported(int Cin, int Cout, int stride) {
    return (Cin == 128 || Cin == 256) && Cout >= 256 && Cout % 256 == 0 && (stride == 1 || stride == 2) ? 1 : 0;
}

extern "C" int sc2_conv1x1_stream_fwd(const void *x, const void *w_frag, const float *bias, const void *residual, void *y,
                                      int N, int H, int W, int Cin, int Cout, int stride, int relu, void *stream) {
    SC2_REQUIRE(x && w_frag && bias && y, SC2_ERR_INVALID_ARG, "conv1x1_stream: null argument");
    SC2_REQUIRE(N > 0 && H > 0 && W > 0, SC2_ERR_INVALID_ARG, "conv1x1_stream: non-positive dimension");
    SC2_REQUIRE(sc2_conv1x1_stream_supported(Cin, Cout, stride), SC2_ERR_UNSUPPORTED,
                "conv1x1_stream: needs Cin in {128, 256}, Cout %% 256 == 0, stride 1 or 2 (got %d -> %d, stride %d)", Cin,
                Cout, stride);
    StreamArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w_frag);
    a.bias = bias;
    a.res = static_cast<const uint16_t *>(residual);
    a.y = static_cast<uint16_t *>(y);
    a.H = H; a.W = W;
    a.OH = (H - 1) / stride + 1;
    a.OW = (W - 1) / stride + 1;
    a.OHW = a.OH * a.OW;
    const long long M = (long long)N * a.OHW;
    SC2_REQUIRE(M < 0x7FFFFFFFLL - 256, SC2_ERR_UNSUPPORTED, "conv1x1_stream: N*OH*OW = %lld exceeds 2^31", M);
    a.M = (int)M; a.Cout = Cout; a.stride = stride; a.relu = relu ? 1 : 0;
    a.n_chunks = Cout / BNC;
    const long long units = ((M + BM - 1) / BM) * a.n_chunks;
    SC2_REQUIRE(units < 0x7FFFFFFFLL - 1024, SC2_ERR_UNSUPPORTED, "conv1x1_stream: too many units");
    a.n_units = (int)units;
    a.unit_ctr = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return Cin == 128 ? launch_stream<128>(a, s) : launch_stream<256>(a, s);
}
