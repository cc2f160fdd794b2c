// GDN1 / inverse GDN1 over 96 channels for training, forward and the whole backward in one launch each (gfx950) -- the first encoder
// normalisation (sc2bench/models/layer.py:476-477: GDN1(96) behind Conv2d(3 -> 96, k5, s2) at 112 x 112; its backward is reached through
// loss.backward() in script/task/image_classification.py:79).  Same quantities as gdn512_rows.hip:
//
//   forward   y = x / (beta + gamma |x|)            |  x * (...)  (inverse)
//   backward  n = beta + gamma |x|;  dd = g / n, dn = -dd x / n  (dd = g n, dn = g x);  dx = dd + sign(x) (gamma^T dn);  d_norm = dn out
//
// Why a kernel of its own: K = 96 is three k-steps -- the two 96 x 96 GEMMs are 0.12 TFLOP against 2.5 GB of tensors (616 MB each at
// 256 x 112 x 112): the op is a streaming pass, and as three launches of the tile kernel (forward 0.48 ms; backward 0.62 + 0.48 ms) it
// moves the tensors three and nine times.  Here every WAVE is an independent worker on strips of 32 pixels x all 96 channels: it pulls
// its strip of x (and g) into its own 7 KB LDS areas with direct-to-LDS loads, runs the GEMM(s) on them with the (tiny) gamma fragments
// read from a workgroup-shared LDS copy, applies the element-wise halves on the accumulators, writes the results back into the same
// areas and streams them out with 16-byte stores.  No barrier after the prologue, no hand-counted waits: eight waves per CU at
// different points of their strips keep the memory system busy (14 KB in flight per wave).
//
// LDS rows are padded to 13 sixteen-byte chunks (208 B): the sixteen pixel rows of an MFMA operand read then start in sixteen different
// groups of four banks (conflict-free), where the natural 192-byte pitch folds them onto four.  A direct-to-LDS instruction fills 1 KB =
// chunk positions 64 p .. 64 p + 63 of the strip; position q holds chunk q % 13 of row q / 13 (the thirteenth chunk and the positions
// past row 31 are fetched out of range: zeros).
//
// sign(0) = 0 (torch.abs's gradient): the direct term dd of EVERY element is parked (bf16) in the g area once g has been consumed; the
// second GEMM accumulates on top of sign(x) dd (0 where x == 0), and the final pass keeps the parked value where x == 0.
#include <stdlib.h>

#include <type_traits>

#include "sc2_common.h"

namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ uint32_t pack2(f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t r, lds_ptr_t dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ void buf_store16(buf_rsrc_t r, uint32_t voff, uint4 v) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, r, (int)voff, 0, 0);
}
#else   // host pass: stand-ins (see conv_igemm_impl.h)
typedef int buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const uint16_t *, uint32_t) { return 0; }
__device__ __forceinline__ void buf_load_lds16(buf_rsrc_t, lds_ptr_t, uint32_t, uint32_t) {}
__device__ __forceinline__ void buf_store16(buf_rsrc_t, uint32_t, uint4) {}
#endif

struct StripArgs {
    const uint16_t *__restrict__ x;      // bf16 [M, 96]
    const uint16_t *__restrict__ gy;     // bf16 [M, 96]             (backward)
    const uint16_t *__restrict__ g1;     // gamma, fragment-major [6 channel tiles][3 k-steps][64 lanes][8]  (hip.pack_weight_fragments)
    const uint16_t *__restrict__ g2;     // gamma^T, the same packing   (backward)
    const float *__restrict__ beta;      // f32 [96]
    uint16_t *__restrict__ out;          // forward: y; backward: dx
    uint16_t *__restrict__ dn;           // backward: d_norm
    float *__restrict__ d_beta;          // backward: column sums of d_norm, accumulated here (zeroed by the caller), or null
    int M, n_strips;
    unsigned bytes;                      // M * 192
};

constexpr int CH = 96, NT = CH / 16, KS = CH / 32, SP = 32, MT = SP / 16, WAVES = 8;
constexpr int ROWB = CH * 2, PITCH = 208, CPR = 13, PIECES = 7;          // 32 rows x 13 chunks = 416 positions <= 7 x 64
constexpr int AREA = PIECES * 1024;                                       // one strip image
constexpr int GAMMA_BYTES = NT * KS * 1024;                               // 18 KB of fragments
constexpr int BETA_OFF = 2 * GAMMA_BYTES, BSUM_OFF = BETA_OFF + 512, AREAS_OFF = BSUM_OFF + 512;   // gamma | gamma^T | beta | d_beta sums | per wave: x area, g area
constexpr int LDS_BYTES = AREAS_OFF + WAVES * 2 * AREA;
constexpr uint32_t OOB = 0x80000000u;
static_assert(SP * CPR <= PIECES * 64 && LDS_BYTES <= 160 * 1024, "strip image, LDS");

// MODE 0: forward; 1: backward
template <int MODE, bool INVERSE>
__global__ __launch_bounds__(WAVES * 64, 2) void gdn96_strips_kernel(const StripArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- workgroup-shared operands: gamma (and gamma^T) fragments, beta
    for (int q = tid; q < GAMMA_BYTES / 16; q += WAVES * 64) {
        reinterpret_cast<uint4 *>(smem)[q] = reinterpret_cast<const uint4 *>(p.g1)[q];
        if (MODE == 1) reinterpret_cast<uint4 *>(smem + GAMMA_BYTES)[q] = reinterpret_cast<const uint4 *>(p.g2)[q];
    }
    if (tid < CH) reinterpret_cast<float *>(smem + BETA_OFF)[tid] = p.beta[tid];
    if (tid < CH) reinterpret_cast<float *>(smem + BSUM_OFF)[tid] = 0.f;
    __syncthreads();   // the only barrier: from here on every wave works alone
    const float *beta_s = reinterpret_cast<const float *>(smem + BETA_OFF);
    unsigned char *xs = smem + AREAS_OFF + wave * (2 * AREA);
    unsigned char *gs = xs + AREA;

    // ---- this lane's seven chunk positions of a strip: position q = 64 p + lane -> (row q / 13, chunk q % 13)
    uint32_t src[PIECES];    // byte offset inside the strip's 32 x 192 bytes, or OOB (padding chunk / past row 31)
    int prow[PIECES];
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) {
        const int q = 64 * pc + lane;
        const int row = q / CPR, c = q - row * CPR;
        prow[pc] = row;
        src[pc] = (row < SP && c < CPR - 1) ? (uint32_t)(row * ROWB + c * 16) : OOB;
    }
    const buf_rsrc_t rs_x = make_rsrc(p.x, p.bytes);
    const buf_rsrc_t rs_g = make_rsrc(MODE == 1 ? p.gy : p.x, p.bytes);
    const buf_rsrc_t rs_out = make_rsrc(p.out, p.bytes);
    const buf_rsrc_t rs_dn = make_rsrc(MODE == 1 ? p.dn : p.out, p.bytes);

    // one GEMM over an area: acc[i][j] += W[tile j][k] * area[row 16 i + frow][k]   (weights as the MFMA A operand: a lane ends up with
    // channels 16 j + 4 fq .. + 3 of pixel 16 i + frow)
    auto gemm = [&](const unsigned char *area, const unsigned char *wfrag, f32x4_t (&acc)[MT][NT], auto abs_c) {
        constexpr bool ABS = decltype(abs_c)::value;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                uint4 v = *reinterpret_cast<const uint4 *>(area + (i * 16 + frow) * PITCH + (ks * 4 + fq) * 16);
                if (ABS) { v.x &= 0x7FFF7FFFu; v.y &= 0x7FFF7FFFu; v.z &= 0x7FFF7FFFu; v.w &= 0x7FFF7FFFu; }
                af[i] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(wfrag + (j * KS + ks) * 1024 + lane * 16));
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[i], acc[i][j], 0, 0, 0);
            }
        }
    };
    // an area -> global: the strip's valid chunks, 16 bytes each (position q of the area is bytes [16 q, 16 q + 16))
    auto stream_out = [&](const unsigned char *area, buf_rsrc_t rs, uint32_t base, int rows_valid) {
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
            const uint4 v = *reinterpret_cast<const uint4 *>(area + pc * 1024 + lane * 16);
            buf_store16(rs, (src[pc] != OOB && prow[pc] < rows_valid) ? base + src[pc] : OOB, v);
        }
    };

    [[maybe_unused]] float bsum[NT][4];   // backward: this lane's running column sums of d_norm (d_beta)
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[j][e] = 0.f;
    }
    const int stride = (int)gridDim.x * WAVES;
    for (int strip = (int)blockIdx.x * WAVES + wave; strip < p.n_strips; strip += stride) {
        const int pix0 = strip * SP;
        const int rows_valid = p.M - pix0 < SP ? p.M - pix0 : SP;
        const uint32_t base = (uint32_t)pix0 * (uint32_t)ROWB;
        // ---------------------------------------------------------------- strip of x (and of g) -> this wave's areas
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
            const uint32_t vo = (src[pc] != OOB && prow[pc] < rows_valid) ? base + src[pc] : OOB;
            buf_load_lds16(rs_x, (lds_ptr_t)(xs + pc * 1024), vo, 0u);
            if (MODE == 1) buf_load_lds16(rs_g, (lds_ptr_t)(gs + pc * 1024), vo, 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also: the previous strip's output stores)

        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        gemm(xs, smem, acc, std::true_type{});             // acc = gamma |x|

        if (MODE == 0) {
            // ------------------------------------------------------------ y = x / (beta + norm)   (x * (...)), in place
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 b4 = *reinterpret_cast<const float4 *>(beta_s + j * 16 + fq * 4);
                const f32x2_t b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    unsigned char *slot = xs + (i * 16 + frow) * PITCH + (j * 16 + fq * 4) * 2;
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
                }
            }
            stream_out(xs, rs_out, base, rows_valid);
            continue;
        }

        // ---------------------------------------------------------------- backward, first half: dn -> x area, dd parked in the g area,
        // sign(x) dd -> acc
        uint32_t zmask[2] = {0u, 0u}, smask[2] = {0u, 0u};   // bit (i * NT + j) * 4 + e: x == 0 / x < 0   (48 bits)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4 b4 = *reinterpret_cast<const float4 *>(beta_s + j * 16 + fq * 4);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int so = (i * 16 + frow) * PITCH + (j * 16 + fq * 4) * 2;
                const uint2 xr = *reinterpret_cast<const uint2 *>(xs + so);
                const uint2 gr = *reinterpret_cast<const uint2 *>(gs + so);
                const uint32_t xb[4] = {xr.x << 16, xr.x & 0xFFFF0000u, xr.y << 16, xr.y & 0xFFFF0000u};
                const float gv[4] = {__builtin_bit_cast(float, gr.x << 16), __builtin_bit_cast(float, gr.x & 0xFFFF0000u),
                                     __builtin_bit_cast(float, gr.y << 16), __builtin_bit_cast(float, gr.y & 0xFFFF0000u)};
                float dnv[4], ddv[4], sdd[4];
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
                    const uint32_t nz = (uint32_t)((int32_t)(0u - (xb[e] & 0x7FFFFFFFu)) >> 31);   // all ones unless x is +-0
                    const int bit = ((i * NT + j) & 7) * 4 + e;
                    zmask[(i * NT + j) >> 3] |= (~nz & 1u) << bit;
                    smask[(i * NT + j) >> 3] |= (xb[e] >> 31) << bit;
                    sdd[e] = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, ddv[e]) ^ (xb[e] & 0x80000000u)) & nz);
                }
                acc[i][j] = f32x4_t{sdd[0], sdd[1], sdd[2], sdd[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[j][e] += dnv[e];
                *reinterpret_cast<uint2 *>(xs + so) = make_uint2(pack2(f32x2_t{dnv[0], dnv[1]}), pack2(f32x2_t{dnv[2], dnv[3]}));
                *reinterpret_cast<uint2 *>(gs + so) = make_uint2(pack2(f32x2_t{ddv[0], ddv[1]}), pack2(f32x2_t{ddv[2], ddv[3]}));
            }
        }
        stream_out(xs, rs_dn, base, rows_valid);            // d_norm
        gemm(xs, smem + GAMMA_BYTES, acc, std::false_type{});   // acc = sign(x) dd + gamma^T dn

        // ---------------------------------------------------------------- dx = sign(x) acc; where x == 0: the parked dd stays
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int so = (i * 16 + frow) * PITCH + (j * 16 + fq * 4) * 2;
                const uint2 parked = *reinterpret_cast<const uint2 *>(gs + so);
                const uint32_t pk[4] = {parked.x << 16, parked.x & 0xFFFF0000u, parked.y << 16, parked.y & 0xFFFF0000u};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = ((i * NT + j) & 7) * 4 + e;
                    const uint32_t sb = ((smask[(i * NT + j) >> 3] >> bit) & 1u) << 31;
                    const uint32_t zero = 0u - ((zmask[(i * NT + j) >> 3] >> bit) & 1u);          // all ones where x == 0
                    const float a = acc[i][j][e];   // (a copy: __builtin_bit_cast of the vector-element lvalue itself reads element 0)
                    const uint32_t v = __builtin_bit_cast(uint32_t, a) ^ sb;
                    o[e] = __builtin_bit_cast(float, (v & ~zero) | (pk[e] & zero));
                }
                *reinterpret_cast<uint2 *>(gs + so) = make_uint2(pack2(f32x2_t{o[0], o[1]}), pack2(f32x2_t{o[2], o[3]}));
            }
        }
        stream_out(gs, rs_out, base, rows_valid);           // dx
    }
    if (MODE == 1 && p.d_beta) {
        // d_beta: the sixteen pixel lanes of a channel quad, the workgroup's waves through LDS, then one atomic per channel
        float *bs = reinterpret_cast<float *>(smem + BSUM_OFF);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bsum[j][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                if (frow == 0) atomicAdd(bs + j * 16 + fq * 4 + e, v);
            }
        __syncthreads();
        if (tid < CH) atomicAdd(p.d_beta + tid, bs[tid]);
    }
}

int g_cus_strips = 0;

template <int MODE, bool INVERSE>
int launch_strips(const StripArgs &a, hipStream_t s) {
    static bool attr_set_dev[SC2_MAX_DEVICES] = {};
    bool &attr_set = attr_set_dev[sc2_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gdn96_strips_kernel<MODE, INVERSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        attr_set = true;
    }
    if (g_cus_strips == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_cus_strips = n;
    }
    const int wgs = (a.n_strips + WAVES - 1) / WAVES;
    const int grid = wgs < g_cus_strips ? wgs : g_cus_strips;     // one 8-wave workgroup per CU (148 KB of LDS)
    hipLaunchKernelGGL((gdn96_strips_kernel<MODE, INVERSE>), dim3(grid), dim3(WAVES * 64), LDS_BYTES, s, a);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

}  // namespace

// (called by sc2_gdn1_rows_fwd / _bwd of gdn512_rows.hip for C == 96: one entry point per operation, whatever the channel count)
int sc2_gdn96_strips(int mode, const void *x, const void *gy, const void *g1, const void *g2, const float *beta, void *out, void *dn,
                     float *d_beta, long long M, int inverse, hipStream_t s) {
    SC2_REQUIRE(M > 0 && M * ROWB < 0x7FF00000LL, SC2_ERR_UNSUPPORTED, "gdn1_rows: %lld pixels x 96 channels exceed 2 GB (32-bit buffer offsets)", M);
    StripArgs a;
    a.x = static_cast<const uint16_t *>(x);
    a.gy = static_cast<const uint16_t *>(gy);
    a.g1 = static_cast<const uint16_t *>(g1);
    a.g2 = static_cast<const uint16_t *>(g2);
    a.beta = beta;
    a.out = static_cast<uint16_t *>(out);
    a.dn = static_cast<uint16_t *>(dn);
    a.d_beta = d_beta;
    a.M = (int)M;
    a.n_strips = (int)((M + SP - 1) / SP);
    a.bytes = (unsigned)(M * ROWB);
    if (mode == 0) return inverse ? launch_strips<0, true>(a, s) : launch_strips<0, false>(a, s);
    return inverse ? launch_strips<1, true>(a, s) : launch_strips<1, false>(a, s);
}
