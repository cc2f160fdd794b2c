// Batched rANS coder for gfx950, bit-exact to CompressAI's RansEncoder.encode_with_indexes /
// RansDecoder.decode_with_indexes (reached from sc2bench/models/layer.py:506 and :520):
// 64-bit state, 32-bit word renormalisation, 16-bit probability precision, 4-bit bypass escape
// for out-of-table values, one independent stream per image.
//
// rANS is a serial state machine per stream, so the parallel axis is the batch: one lane per
// stream, 64 streams per wave.  All lanes of a wave sit at the same symbol position, so table
// rows are (nearly) wave-uniform and the quantised CDFs live in LDS.
//   * encoder: walks the symbols back to front (the order in which upstream's flush() pops its
//     symbol stack), emits 32-bit words from the END of the stream's row towards its start, then
//     the two state words; the stream is therefore end-aligned in its row (out_offset tells where).
//     x / freq uses an exact double-precision reciprocal division (2 fma-corrected steps) instead
//     of a 64-bit integer division.
//   * decoder: mirrors Rans64DecGet/Advance; the symbol search is an upper-bound binary search
//     over the (strictly increasing) CDF row, identical in result to upstream's linear find_if.
#include "sc2_common.h"

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassPrecision = 4;
constexpr int kMaxBypassVal = (1 << kBypassPrecision) - 1;
constexpr unsigned long long kRansL = 1ull << 31;
constexpr int kMaxLdsEntries = 6144;  // cdf entries kept in LDS (24 KB i32 + 48 KB f64 reciprocals)

struct RansArgs {
    const int32_t *symbols;   // encode: in   decode: out
    const int32_t *indexes;   // nullable
    long long index_div;
    int n_streams;
    long long n_sym;
    const int32_t *cdfs;
    int n_cdfs, cdf_stride;
    const int32_t *cdf_sizes;
    const int32_t *offsets;
    uint8_t *buf;             // encode: out  decode: in
    long long stride;
    int32_t *io_offset;       // encode: out  decode: in
    int32_t *io_nbytes;       // encode: out  decode: in
    int32_t *status;
    int32_t *symbols_out;
};

// exact floor(x / f) and x mod f for x < 2^63, 1 <= f < 2^16, with rcp = 1.0 / f (correctly rounded).
__device__ __forceinline__ void divmod_u64(unsigned long long x, unsigned f, double rcp, unsigned long long &q,
                                           unsigned &r) {
    const double df = (double)f;
    const double dxh = (double)(unsigned)(x >> 32);
    double q1 = floor(dxh * rcp);
    double r1 = fma(-q1, df, dxh);
    if (r1 >= df) { q1 += 1.0; r1 -= df; }
    if (r1 < 0.0) { q1 -= 1.0; r1 += df; }
    const double num = fma(r1, 4294967296.0, (double)(unsigned)x);  // < 2^48, exact
    double q0 = floor(num * rcp);
    double r0 = fma(-q0, df, num);
    if (r0 >= df) { q0 += 1.0; r0 -= df; }
    if (r0 < 0.0) { q0 -= 1.0; r0 += df; }
    q = ((unsigned long long)(unsigned)q1 << 32) | (unsigned long long)(unsigned)q0;
    r = (unsigned)r0;
}

struct EncState {
    unsigned long long x;
    uint32_t *ptr;    // next free word is ptr[-1]
    uint32_t *limit;  // lowest address that still leaves room for the 2 flush words
    int overflow;
};

__device__ __forceinline__ void enc_emit(EncState &s) {
    if (s.ptr > s.limit) {
        s.ptr -= 1;
        *s.ptr = (uint32_t)s.x;
    } else {
        s.overflow = 1;
    }
    s.x >>= 32;
}

__device__ __forceinline__ void enc_put(EncState &s, unsigned start, unsigned freq, double rcp) {
    // x_max = ((RANS64_L >> 16) << 32) * freq = freq << 47
    if ((s.x >> 47) >= (unsigned long long)freq) enc_emit(s);
    unsigned long long q;
    unsigned r;
    divmod_u64(s.x, freq, rcp, q, r);
    s.x = (q << kPrecision) + r + start;
}

__device__ __forceinline__ void enc_put_bits(EncState &s, unsigned val) {
    // freq = 1 << (16 - 4); x_max = 2^59
    if ((s.x >> 59) != 0ull) enc_emit(s);
    s.x = (s.x << kBypassPrecision) | val;
}

template <bool LDS_TABLES>
__global__ __launch_bounds__(64) void rans_encode_kernel(const RansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t *l_cdf = reinterpret_cast<int32_t *>(smem);
    const int n_entries = a.n_cdfs * a.cdf_stride;
    double *l_rcp = reinterpret_cast<double *>(smem + ((n_entries * 4 + 15) / 16) * 16);
    if (LDS_TABLES) {
        for (int i = threadIdx.x; i < n_entries; i += 64) {
            l_cdf[i] = a.cdfs[i];
            const int row = i / a.cdf_stride, col = i - row * a.cdf_stride;
            double rc = 0.0;
            if (col + 1 < a.cdf_stride) {
                const int f = a.cdfs[i + 1] - a.cdfs[i];
                if (f > 0) rc = 1.0 / (double)f;
            }
            l_rcp[i] = rc;
        }
        __syncthreads();
    }
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= a.n_streams) return;

    const int32_t *sym = a.symbols + (long long)s * a.n_sym;
    const int32_t *idxp = a.indexes ? a.indexes + (long long)s * a.n_sym : nullptr;
    uint32_t *row = reinterpret_cast<uint32_t *>(a.buf + (long long)s * a.stride);
    const long long row_words = a.stride / 4;

    EncState st;
    st.x = kRansL;
    st.ptr = row + row_words;
    st.limit = row + 2;
    st.overflow = 0;

    for (long long i = a.n_sym - 1; i >= 0; --i) {
        const int idx = idxp ? idxp[i] : (int)(i / a.index_div);
        const int max_value = a.cdf_sizes[idx] - 2;
        int value = sym[i] - a.offsets[idx];
        unsigned raw_val = 0;
        if (value < 0) {
            raw_val = (unsigned)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw_val = (unsigned)(2 * (value - max_value));
            value = max_value;
        }
        if (value == max_value) {
            // upstream pushes [symbol, count nibbles (15,..,15,rem), raw nibbles j=0..n-1]; flush pops in reverse.
            int n_bypass = 0;
            while ((raw_val >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
            for (int j = n_bypass - 1; j >= 0; --j) enc_put_bits(st, (raw_val >> (j * kBypassPrecision)) & kMaxBypassVal);
            const int n15 = n_bypass / kMaxBypassVal, rem = n_bypass - n15 * kMaxBypassVal;
            enc_put_bits(st, (unsigned)rem);
            for (int t = 0; t < n15; ++t) enc_put_bits(st, kMaxBypassVal);
        }
        const int e = idx * a.cdf_stride + value;
        unsigned start, freq;
        double rcp;
        if (LDS_TABLES) {
            start = (unsigned)l_cdf[e];
            freq = (unsigned)l_cdf[e + 1] - start;
            rcp = l_rcp[e];
        } else {
            start = (unsigned)a.cdfs[e];
            freq = (unsigned)a.cdfs[e + 1] - start;
            rcp = 1.0 / (double)(freq & 0xFFFFu);
        }
        // upstream stores start and range as uint16_t
        enc_put(st, start & 0xFFFFu, freq & 0xFFFFu, rcp);
    }
    // Rans64EncFlush
    st.ptr -= 2;
    st.ptr[0] = (uint32_t)(st.x);
    st.ptr[1] = (uint32_t)(st.x >> 32);
    a.io_offset[s] = (int32_t)((st.ptr - row) * 4);
    a.io_nbytes[s] = (int32_t)((row + row_words - st.ptr) * 4);
    a.status[s] = st.overflow;
}

struct DecState {
    unsigned long long x;
    const uint32_t *ptr;
    const uint32_t *end;
};

__device__ __forceinline__ void dec_renorm(DecState &d) {
    if (d.x < kRansL) {
        const uint32_t w = d.ptr < d.end ? *d.ptr : 0u;
        d.ptr += 1;
        d.x = (d.x << 32) | w;
    }
}
__device__ __forceinline__ unsigned dec_get_bits(DecState &d) {
    const unsigned val = (unsigned)(d.x & kMaxBypassVal);
    d.x >>= kBypassPrecision;
    dec_renorm(d);
    return val;
}

template <bool LDS_TABLES>
__global__ __launch_bounds__(64) void rans_decode_kernel(const RansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t *l_cdf = reinterpret_cast<int32_t *>(smem);
    const int n_entries = a.n_cdfs * a.cdf_stride;
    if (LDS_TABLES) {
        for (int i = threadIdx.x; i < n_entries; i += 64) l_cdf[i] = a.cdfs[i];
        __syncthreads();
    }
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= a.n_streams) return;
    const int32_t *tab = LDS_TABLES ? l_cdf : a.cdfs;

    const int32_t *idxp = a.indexes ? a.indexes + (long long)s * a.n_sym : nullptr;
    int32_t *out = a.symbols_out + (long long)s * a.n_sym;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.buf + (long long)s * a.stride + a.io_offset[s]);
    const int n_words = a.io_nbytes[s] / 4;
    DecState d;
    d.end = w + n_words;
    d.x = (unsigned long long)(n_words > 0 ? w[0] : 0u) | ((unsigned long long)(n_words > 1 ? w[1] : 0u) << 32);
    d.ptr = w + 2;

    for (long long i = 0; i < a.n_sym; ++i) {
        const int idx = idxp ? idxp[i] : (int)(i / a.index_div);
        const int32_t *cdf = tab + idx * a.cdf_stride;
        const int size = a.cdf_sizes[idx];
        const int max_value = size - 2;
        const int offset = a.offsets[idx];
        const unsigned cum_freq = (unsigned)(d.x & 0xFFFFu);
        // first k in [0, size) with cdf[k] > cum_freq  (upper bound); s = k - 1
        int lo = 0, hi = size;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((unsigned)cdf[mid] > cum_freq) hi = mid; else lo = mid + 1;
        }
        const int sidx = lo - 1;
        const unsigned start = (unsigned)cdf[sidx];
        const unsigned freq = (unsigned)cdf[sidx + 1] - start;
        d.x = (unsigned long long)freq * (d.x >> kPrecision) + cum_freq - start;
        dec_renorm(d);
        int value = sidx;
        if (value == max_value) {
            int val = (int)dec_get_bits(d);
            int n_bypass = val;
            while (val == kMaxBypassVal) {
                val = (int)dec_get_bits(d);
                n_bypass += val;
            }
            int raw_val = 0;
            for (int j = 0; j < n_bypass; ++j) {
                val = (int)dec_get_bits(d);
                raw_val |= val << (j * kBypassPrecision);
            }
            value = raw_val >> 1;
            if (raw_val & 1) value = -value - 1;
            else value += max_value;
        }
        out[i] = value + offset;
    }
    a.status[s] = 0;
}

int check_common(const int32_t *indexes, long long index_div, int n_streams, long long n_sym, const int32_t *cdfs,
                 int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets) {
    SC2_REQUIRE(cdfs && cdf_sizes && offsets, SC2_ERR_INVALID_ARG, "rans: Uninitialized CDFs. Run update() first");
    SC2_REQUIRE(n_streams > 0 && n_sym >= 0 && n_cdfs > 0 && cdf_stride >= 3, SC2_ERR_INVALID_ARG,
                "rans: bad sizes n_streams=%d n_sym=%lld n_cdfs=%d cdf_stride=%d", n_streams, n_sym, n_cdfs,
                cdf_stride);
    if (!indexes) SC2_REQUIRE(index_div > 0, SC2_ERR_INVALID_ARG, "rans: index_div must be positive");
    if (!indexes && n_sym > 0)
        SC2_REQUIRE((n_sym - 1) / index_div < n_cdfs, SC2_ERR_INVALID_ARG,
                    "rans: implicit index %lld out of range (%d CDF rows)", (n_sym - 1) / index_div, n_cdfs);
    return SC2_OK;
}

}  // namespace

extern "C" int64_t sc2_rans_max_bytes(int64_t n_sym) {
    // worst case per symbol: 16 bits (freq 1) + escape: count nibble + 8 raw nibbles = 52 bits; + 8 flush bytes.
    if (n_sym < 0) return 0;
    const int64_t words = (n_sym * 52 + 31) / 32 + 4;
    return words * 4;
}

extern "C" int sc2_rans_encode_batch(const int32_t *symbols, const int32_t *indexes, int64_t index_div, int n_streams,
                                     int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                     const int32_t *cdf_sizes, const int32_t *offsets, uint8_t *out,
                                     int64_t out_stride, int32_t *out_offset, int32_t *out_nbytes, int32_t *status,
                                     void *stream) {
    int rc = check_common(indexes, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets);
    if (rc != SC2_OK) return rc;
    SC2_REQUIRE((symbols || n_sym == 0) && out && out_offset && out_nbytes && status, SC2_ERR_INVALID_ARG,
                "rans_encode: null argument");
    SC2_REQUIRE(out_stride >= 16 && out_stride % 4 == 0 && out_stride < (1ll << 31), SC2_ERR_INVALID_ARG,
                "rans_encode: out_stride %lld must be a multiple of 4 in [16, 2^31)", (long long)out_stride);
    RansArgs a;
    a.symbols = symbols; a.indexes = indexes; a.index_div = index_div > 0 ? index_div : 1;
    a.n_streams = n_streams; a.n_sym = n_sym;
    a.cdfs = cdfs; a.n_cdfs = n_cdfs; a.cdf_stride = cdf_stride; a.cdf_sizes = cdf_sizes; a.offsets = offsets;
    a.buf = out; a.stride = out_stride; a.io_offset = out_offset; a.io_nbytes = out_nbytes; a.status = status;
    a.symbols_out = nullptr;
    const int grid = (n_streams + 63) / 64;
    const int n_entries = n_cdfs * cdf_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_entries <= kMaxLdsEntries) {
        const size_t lds = ((size_t)(n_entries * 4 + 15) / 16) * 16 + (size_t)n_entries * 8;
        hipLaunchKernelGGL(rans_encode_kernel<true>, dim3(grid), dim3(64), lds, s, a);
    } else {
        hipLaunchKernelGGL(rans_encode_kernel<false>, dim3(grid), dim3(64), 0, s, a);
    }
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_rans_decode_batch(const uint8_t *in, int64_t in_stride, const int32_t *in_offset,
                                     const int32_t *in_nbytes, const int32_t *indexes, int64_t index_div,
                                     int n_streams, int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                     const int32_t *cdf_sizes, const int32_t *offsets, int32_t *symbols_out,
                                     int32_t *status, void *stream) {
    int rc = check_common(indexes, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets);
    if (rc != SC2_OK) return rc;
    SC2_REQUIRE(in && in_offset && in_nbytes && (symbols_out || n_sym == 0) && status, SC2_ERR_INVALID_ARG,
                "rans_decode: null argument");
    SC2_REQUIRE(in_stride >= 8 && in_stride % 4 == 0, SC2_ERR_INVALID_ARG,
                "rans_decode: in_stride %lld must be a multiple of 4, >= 8", (long long)in_stride);
    RansArgs a;
    a.symbols = nullptr; a.indexes = indexes; a.index_div = index_div > 0 ? index_div : 1;
    a.n_streams = n_streams; a.n_sym = n_sym;
    a.cdfs = cdfs; a.n_cdfs = n_cdfs; a.cdf_stride = cdf_stride; a.cdf_sizes = cdf_sizes; a.offsets = offsets;
    a.buf = const_cast<uint8_t *>(in); a.stride = in_stride;
    a.io_offset = const_cast<int32_t *>(in_offset); a.io_nbytes = const_cast<int32_t *>(in_nbytes);
    a.status = status; a.symbols_out = symbols_out;
    const int grid = (n_streams + 63) / 64;
    const int n_entries = n_cdfs * cdf_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_entries <= kMaxLdsEntries) {
        hipLaunchKernelGGL(rans_decode_kernel<true>, dim3(grid), dim3(64), (size_t)n_entries * 4, s, a);
    } else {
        hipLaunchKernelGGL(rans_decode_kernel<false>, dim3(grid), dim3(64), 0, s, a);
    }
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}
