// Batched rANS coder for gfx950, bit-exact to CompressAI's RansEncoder.encode_with_indexes /
// RansDecoder.decode_with_indexes (reached from sc2bench/models/layer.py:506 and :520):
// 64-bit state, 32-bit word renormalisation, 16-bit probability precision, 4-bit bypass escape
// for out-of-table values, one independent stream per image.
//
// rANS is a serial state machine per stream, so the parallel axis is the batch: one lane per stream,
// 64 streams per wave.  Everything that is NOT on the state's dependency chain is moved out of the serial
// loop into fully parallel passes, and every access of the serial loop is made lane-contiguous:
//
//   encode  = table (parallel: per CDF entry the Alverson reciprocal of its frequency)  ->  prepare (parallel:
//             symbol -> table entry index, transposed to [position][lane] through an LDS tile)  ->  serial (one
//             wave per 64 streams: per 8 positions, 8 coalesced 256-byte loads and 8 LDS table reads are issued up
//             front, then 8 branch-free integer state updates x += bias + (mulhi64(x, rcp) >> shift) * (2^16 - freq);
//             emitted words are staged per lane in LDS and flushed in bursts).  Words are laid out from the END of
//             the stream's row towards its start (the order in which upstream's flush() pops its symbol stack), so
//             a stream is end-aligned in its row.  A conditional branch costs ~60 cycles on a lone wave
//             (tools/micro/chain2.hip), hence the select-based hot path; escapes take a per-chunk slow path.
//   decode  = serial (per CDF row, the wave builds an exact 65536-entry cum_freq -> symbol table in LDS: one
//             ds_read_u16 replaces upstream's linear find_if; stream words come from a per-lane LDS ring; 8-symbol
//             chunks are speculated branch-free and rolled back if they hit an escape symbol; outputs go to a
//             [position][lane] buffer)  ->  finish (parallel transpose back to [stream][position]).
//   Per-symbol explicit `indexes` (no common row per position) use the generic kernels at the bottom.
#include "sc2_common.h"

#ifndef SC2_RANS_PRIO
#define SC2_RANS_PRIO 3   // wave priority of the serial coder kernels (A/B: -DSC2_RANS_PRIO=0, tools/build_variant.sh)
#endif

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassPrecision = 4;
constexpr int kMaxBypassVal = (1 << kBypassPrecision) - 1;
constexpr unsigned long long kRansL = 1ull << 31;
constexpr int kEncLdsEntries = 4096;   // encoder table entries (16 B each) kept in LDS
constexpr int kMaxRowLds = 4096;       // longest CDF row the LUT decoder keeps in LDS

struct RansArgs {
    const int32_t *symbols;   // encode: in
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
    int32_t *symbols_out;     // decode: out
    uint32_t *ws;             // [n_blocks][n_sym][64]
};

// ------------------------------------------------------------------------------------------------ encode
// Per-entry encoder constants (Alverson reciprocal, as ryg_rans' Rans64EncSymbol): with
//   q = mulhi64(x, rcp) >> shift   (exact floor(x / freq) for x < 2^63)
// the rANS update ((x / freq) << 16) + (x % freq) + start equals x + bias + q * (65536 - freq):
// no division and no remainder on the state's dependency chain.
struct __attribute__((aligned(16))) EncEntry {
    unsigned long long rcp;
    uint32_t bias;   // start (freq >= 2) or start + 65535 (freq == 1)
    uint32_t cs;     // (65536 - freq) | shift << 16
};

__global__ __launch_bounds__(256) void rans_build_enc_table_kernel(const int32_t *__restrict__ cdfs, int n_entries,
                                                                   int cdf_stride, EncEntry *__restrict__ tab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_entries) return;
    EncEntry en;
    en.rcp = ~0ull; en.bias = 65535u; en.cs = 65535u;   // harmless filler for the last column / empty cells
    if (i % cdf_stride + 1 < cdf_stride) {
        const uint32_t start = (uint32_t)cdfs[i] & 0xFFFFu;                                 // uint16_t upstream
        const uint32_t freq = ((uint32_t)cdfs[i + 1] - (uint32_t)cdfs[i]) & 0xFFFFu;
        if (freq >= 2) {
            uint32_t shift = 0;
            while (freq > (1u << shift)) ++shift;
            // ((1 << (shift + 63)) + freq - 1) / freq by two 64-bit divides
            unsigned long long x0 = freq - 1, x1 = 1ull << (shift + 31);
            const unsigned long long t1 = x1 / freq;
            x0 += (x1 % freq) << 32;
            const unsigned long long t0 = x0 / freq;
            en.rcp = t0 + (t1 << 32);
            en.bias = start;
            en.cs = (65536u - freq) | ((shift - 1) << 16);
        } else if (freq == 1) {
            en.rcp = ~0ull;
            en.bias = start + 65535u;
            en.cs = 65535u;
        }
    }
    tab[i] = en;
}

// parallel pass: ws[blk][i][lane] = table entry index of symbol i of stream blk*64+lane, bit 31 = escape.
__global__ __launch_bounds__(256) void rans_enc_prepare_kernel(const RansArgs a) {
    __shared__ uint32_t tile[64][65];
    const int blk = blockIdx.y;
    const long long i0 = (long long)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int pl = t & 63;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int sl = r * 4 + (t >> 6);
        const int s = blk * 64 + sl;
        const long long i = i0 + pl;
        uint32_t e = 0;
        if (s < a.n_streams && i < a.n_sym) {
            const long long g = (long long)s * a.n_sym + i;
            const int idx = a.indexes ? a.indexes[g] : (int)(i / a.index_div);
            const int max_value = a.cdf_sizes[idx] - 2;
            int value = a.symbols[g] - a.offsets[idx];
            uint32_t esc = 0;
            if (value < 0 || value >= max_value) { value = max_value; esc = 0x80000000u; }
            e = (uint32_t)(idx * a.cdf_stride + value) | esc;
        }
        tile[pl][sl] = e;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int p = r * 4 + (t >> 6);
        const long long i = i0 + p;
        if (i < a.n_sym) a.ws[((long long)blk * a.n_sym + i) * 64 + pl] = tile[p][pl];
    }
}

constexpr int kStage = 48;   // emitted words staged per lane in LDS between flushes (+1 dummy slot)

template <bool LDS_TABLES>
__global__ __launch_bounds__(64) void rans_enc_serial_kernel(const RansArgs a, const EncEntry *__restrict__ gtab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *stg = reinterpret_cast<uint32_t *>(smem);                                   // [kStage + 1][64]
    EncEntry *ltab = reinterpret_cast<EncEntry *>(smem + (kStage + 1) * 64 * 4);
    const int n_entries = a.n_cdfs * a.cdf_stride;
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(SC2_RANS_PRIO);   // a lone serial wave must not queue behind co-resident MFMA workgroups
    if (LDS_TABLES) {
        for (int i = lane; i < n_entries; i += 64) ltab[i] = gtab[i];
        __syncthreads();
    }
    const EncEntry *tab = LDS_TABLES ? ltab : gtab;
    const int blk = blockIdx.x;
    const int s = blk * 64 + lane;
    const bool active = s < a.n_streams;
    const int sc = active ? s : a.n_streams - 1;  // inactive lanes shadow a real stream but never store
    uint32_t *row = reinterpret_cast<uint32_t *>(a.buf + (long long)sc * a.stride);
    const long long row_words = a.stride / 4;
    const uint32_t *wsb = a.ws + (long long)blk * a.n_sym * 64 + lane;

    unsigned long long x = kRansL;
    uint32_t *ptr = row + row_words;                 // next free word is ptr[-1]
    uint32_t *const limit = active ? row + 2 : row + row_words;   // keeps room for the 2 flush words
    int overflow = 0;
    int cnt = 0;                                     // words staged in LDS for this lane
    uint32_t *const stl = stg + lane;

    auto flush = [&]() {
        long long avail = ptr - limit;
        int n_store = cnt;
        if ((long long)cnt > avail) { n_store = (int)avail; overflow |= 1; }   // (keeps bit 1 = clamped symbol)
        // four words per store where a lane has them (every lane writes ITS row: a 4-byte store per lane is 64 separate
        // requests per wave-instruction); dword-aligned dwordx4, words in descending address order as emitted
        struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
        const int n4 = n_store & ~3;
        for (int j = 0; __any(j < n4); j += 4) {
            if (j < n4) {
                Words4 t;
                t.v[3] = stl[j * 64]; t.v[2] = stl[(j + 1) * 64]; t.v[1] = stl[(j + 2) * 64]; t.v[0] = stl[(j + 3) * 64];
                *reinterpret_cast<Words4 *>(ptr - 4 - j) = t;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (n4 + r < n_store) ptr[-1 - (n4 + r)] = stl[(n4 + r) * 64];
        ptr -= n_store;
        cnt = 0;
    };
    // branch-free table symbol (the common case): selects are written as mask arithmetic so that the compiler
    // keeps them as VALU bit operations instead of exec-mask control flow (VALU->SALU round trips cost more
    // than the arithmetic they guard on a lone wave).
    auto put_fast = [&](const EncEntry en) {
        const uint32_t cmpl = en.cs & 0xFFFFu, shift = en.cs >> 16, freq = 65536u - cmpl;
        const uint32_t xh = (uint32_t)(x >> 32), xl = (uint32_t)x;
        // emit <=> x >= freq << 47 <=> (xh >> 15) >= freq ; m = all-ones when emitting
        const uint32_t m = (uint32_t)((int32_t)(freq - 1u - (xh >> 15)) >> 31);
        const uint32_t slot = (uint32_t)kStage + (((uint32_t)cnt - (uint32_t)kStage) & m);
        stl[slot * 64] = xl;
        cnt -= (int)m;   // m is 0 or -1
        const uint32_t nxl = xl ^ ((xl ^ xh) & m);
        const uint32_t nxh = xh & ~m;
        x = ((unsigned long long)nxh << 32) | nxl;
        const unsigned long long q = __umul64hi(x, en.rcp) >> shift;
        x = x + en.bias + q * cmpl;
    };
    auto emit_slow = [&]() {
        stl[cnt * 64] = (uint32_t)x;
        cnt += 1;
        x >>= 32;
    };
    auto put_bits = [&](unsigned val) {              // bypass: freq = 1 << 12, x_max = 2^59
        if ((x >> 59) != 0ull) emit_slow();
        x = (x << kBypassPrecision) | val;
    };
    // one symbol with the out-of-table (bypass) path; upstream pushes [symbol, count nibbles (15,..,15,rem), raw
    // nibbles j=0..n-1] and flush() pops them in reverse, so the bypass part is coded BEFORE the table symbol.
    auto step_slow = [&](uint32_t e, long long i) {
        if (__any(cnt > kStage - 14)) flush();
        const uint32_t ei = e & 0x7FFFFFFFu;
        if (e >> 31) {
            const int idx = (int)(ei / (uint32_t)a.cdf_stride);
            const int max_value = a.cdf_sizes[idx] - 2;
            // symbol - offset in 64 bits: a latent saturated by eb_symbols (INT_MIN / INT_MAX) must not wrap before the clamp
            const long long v = (long long)a.symbols[(long long)sc * a.n_sym + i] - (long long)a.offsets[idx];
            // |v| >= 2^30 would wrap the 32-bit raw value (a diverged latent saturated by eb_symbols): code the
            // clamped value and report status 2 instead of spinning (a shift by 32 is masked to 0 on gfx950, so
            // upstream's unbounded nibble count would never terminate here)
            int vc = (int)(v < -(1ll << 30) ? -(1ll << 30) : (v > (long long)max_value + (1ll << 30) ? (long long)max_value + (1ll << 30) : v));
            if (v != (long long)vc) overflow |= 2;
            const unsigned raw = vc < 0 ? (unsigned)(-2 * vc - 1) : (unsigned)(2 * (vc - max_value));
            int n_bypass = 0;
            while (n_bypass < 8 && (raw >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
            for (int j = n_bypass - 1; j >= 0; --j) put_bits((raw >> (j * kBypassPrecision)) & kMaxBypassVal);
            const int n15 = n_bypass / kMaxBypassVal, rem = n_bypass - n15 * kMaxBypassVal;
            put_bits((unsigned)rem);
            for (int t = 0; t < n15; ++t) put_bits(kMaxBypassVal);
        }
        put_fast(tab[ei]);
    };

    constexpr int U = 8;
    long long i = a.n_sym - 1;
    for (long long rem = a.n_sym % U; rem > 0; --rem, --i) step_slow(wsb[i * 64], i);   // ragged head
    // The entries of a chunk are loaded TWO chunks ahead, in three register sets that rotate by name (copying one set into
    // another would make the wave wait for it a chunk early): the [position][lane] intermediate of a 2 048-stream launch is
    // 600 MB, it comes from HBM, and one chunk of symbols (~1 us) is shorter than a loaded HBM round trip.
    auto load_chunk = [&](uint32_t (&e)[U], long long i0) {
        if (i0 >= 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = wsb[(i0 - u) * 64];
        }
    };
    auto code_chunk = [&](const uint32_t (&e)[U], long long i0) {
        uint32_t esc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) esc |= e[u];
        if (__any((esc >> 31) != 0)) {
#pragma unroll 1
            for (int u = 0; u < U; ++u) step_slow(e[u], i0 - u);
        } else {
            if (__any(cnt > kStage - U)) flush();
            // (the fence: hipcc otherwise sinks each table read down to its symbol -- `ds_read2_b64; s_waitcnt lgkmcnt(0)` per
            //  symbol in the listing -- and the serial chain pays an LDS round trip per symbol instead of one per chunk)
            EncEntry en[U];
#pragma unroll
            for (int u = 0; u < U; ++u) en[u] = tab[e[u]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) put_fast(en[u]);
        }
    };
    uint32_t e0[U], e1[U], e2[U];
    load_chunk(e0, i);
    load_chunk(e1, i - U);
    while (i >= 0) {
        load_chunk(e2, i - 2 * U);
        code_chunk(e0, i);
        i -= U;
        if (i < 0) break;
        load_chunk(e0, i - 2 * U);
        code_chunk(e1, i);
        i -= U;
        if (i < 0) break;
        load_chunk(e1, i - 2 * U);
        code_chunk(e2, i);
        i -= U;
    }
    flush();
    if (active) {
        // Rans64EncFlush
        ptr -= 2;
        ptr[0] = (uint32_t)(x);
        ptr[1] = (uint32_t)(x >> 32);
        a.io_offset[s] = (int32_t)((ptr - row) * 4);
        a.io_nbytes[s] = (int32_t)((row + row_words - ptr) * 4);
        a.status[s] = overflow;
    }
}

// ------------------------------------------------------------------------------------------------ decode
struct DecState {
    unsigned long long x;
    const uint32_t *ptr;   // next word to prefetch
    const uint32_t *end;
    uint32_t w0, w1;       // the next two words of the stream, already loaded
};

__device__ __forceinline__ uint32_t dec_fetch(const uint32_t *p, const uint32_t *end) { return p < end ? *p : 0u; }

__device__ __forceinline__ void dec_init(DecState &d, const uint32_t *w, int n_words) {
    d.end = w + n_words;
    d.x = (unsigned long long)dec_fetch(w, d.end) | ((unsigned long long)dec_fetch(w + 1, d.end) << 32);
    d.w0 = dec_fetch(w + 2, d.end);
    d.w1 = dec_fetch(w + 3, d.end);
    d.ptr = w + 4;
}
__device__ __forceinline__ void dec_renorm(DecState &d) {
    if (d.x < kRansL) {
        d.x = (d.x << 32) | d.w0;
        d.w0 = d.w1;
        d.w1 = dec_fetch(d.ptr, d.end);
        d.ptr += 1;
    }
}
__device__ __forceinline__ unsigned dec_get_bits(DecState &d) {
    const unsigned val = (unsigned)(d.x & kMaxBypassVal);
    d.x >>= kBypassPrecision;
    dec_renorm(d);
    return val;
}
// Status bit 3: a corrupt, truncated or hostile stream.  A value the encoder escapes is a 32-bit raw value, i.e. at most eight
// nibbles, so its nibble count is ONE count digit <= 8: upstream's `while (val == 15)` continuation loop never runs on a stream
// an encoder produced.  A decoder that trusts it spins for ever on a stream of 0xFF bytes (x = 2^64 - 1: every nibble is 0xF,
// every renormalisation word is all ones); decompress() takes its bytes from the network in split computing, so anything but a
// single digit <= 8 ends the escape here, flags the stream and codes the escape entry itself.  The same bit reports a stream
// whose decoding read words past its end (zeros are supplied there).
constexpr int kStatusCorrupt = 8;
constexpr int kMaxEscapeNibbles = 8;

template <class GetBits>
__device__ __forceinline__ int decode_escape(GetBits &&get_bits, int max_value, int &corrupt) {
    const int n_bypass = (int)get_bits();
    if (n_bypass > kMaxEscapeNibbles) {
        corrupt = kStatusCorrupt;
        return max_value;
    }
    int raw_val = 0;
    for (int j = 0; j < n_bypass; ++j) raw_val |= (int)get_bits() << (j * kBypassPrecision);
    int value = raw_val >> 1;
    if (raw_val & 1) value = -value - 1;
    else value += max_value;
    return value;
}
__device__ __forceinline__ int dec_escape(DecState &d, int max_value, int &corrupt) {
    return decode_escape([&]() { return dec_get_bits(d); }, max_value, corrupt);
}
// words of the stream consumed so far (the two look-ahead words are fetched, not consumed)
__device__ __forceinline__ bool dec_past_end(const DecState &d, const uint32_t *w, int n_words) {
    return (d.ptr - 2) - w > n_words;
}

constexpr int kWin = 32;   // stream words buffered per lane in LDS (ring)

// LUT decoder for implicit indexes (row = position / index_div): all lanes share the CDF row of the current
// position.  Per row the wave builds an exact 65536-entry u16 table cum_freq -> symbol in LDS; the symbol's
// {start, freq} comes from a second small LDS table.  Stream words are staged per lane in an LDS ring that is
// topped up 16 words at a time, so the common path (8 symbols) touches no global loads and has no branches;
// chunks that contain an escape symbol are rolled back and redone on the exact per-symbol path.
// LutT = uint8_t when every CDF row has at most 256 symbols (the entropy bottleneck's tables: 10 - 100): the table is then
// 64 KB instead of 128 KB and TWO decode workgroups share a CU -- a 2 048-stream launch holds 16 CUs for its ~15 ms instead of
// 32 (a CU that hosts a decode wave cannot host a 128 KB workgroup of the MFMA kernels running beside it).
template <typename LutT>
__global__ __launch_bounds__(64) void rans_dec_lut_kernel(const RansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LutT *lut = reinterpret_cast<LutT *>(smem);                                             // [65536]
    uint32_t *win = reinterpret_cast<uint32_t *>(smem + 65536 * sizeof(LutT));              // [kWin][64]
    uint32_t *rowtab = win + kWin * 64;                                                     // start | freq << 16
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    const int s = blk * 64 + lane;
    const bool active = s < a.n_streams;
    const int sc = active ? s : a.n_streams - 1;
    __builtin_amdgcn_s_setprio(SC2_RANS_PRIO);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.buf + (long long)sc * a.stride + a.io_offset[sc]);
    const int n_words = a.io_nbytes[sc] / 4;
    uint32_t *wsb = a.ws + (long long)blk * a.n_sym * 64 + lane;
    uint32_t *const wl = win + lane;

    int corrupt = 0;   // kStatusCorrupt once an escape of this stream is malformed
    int lp = 0;   // words [0, lp) of this lane's stream have been copied into the ring
    int rp = 0;   // next unread word
    // every lane whose ring has >= 16 free slots takes its next 16 words (0 past the end) as four 16-byte loads: a lane's words
    // are consecutive in ITS stream and every lane reads another row, so a 4-byte load per lane is 64 separate 4-byte requests
    // per wave-instruction -- four times the requests of the same bytes as dwordx4 (dword-aligned, which gfx950 allows)
    // ... and one top-up AHEAD: pre[] holds words [lp, lp + 16) of the lane's stream, loaded when the previous top-up ran, so a
    // top-up writes registers to the ring and issues the next loads without waiting for memory
    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
    uint32_t pre[16];
    auto prefetch = [&]() {   // pre[] <- words [lp, lp + 16)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = lp + 4 * q;
            Words4 t;
            if (k + 3 < n_words) {
                t = *reinterpret_cast<const Words4 *>(w + k);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t.v[e] = k + e < n_words ? w[k + e] : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[4 * q + e] = t.v[e];
        }
    };
    prefetch();
    auto top_up = [&]() {
        const bool want = lp - rp <= kWin - 16;
        if (want) {
#pragma unroll
            for (int j = 0; j < 16; ++j) wl[((lp + j) & (kWin - 1)) * 64] = pre[j];
            lp += 16;
            prefetch();
        }
    };
    auto ring = [&](int k) { return wl[(k & (kWin - 1)) * 64]; };
    top_up();
    top_up();
    unsigned long long x = (unsigned long long)ring(0) | ((unsigned long long)ring(1) << 32);
    rp = 2;
    uint32_t wq = ring(rp), wq1 = ring(rp + 1);   // the next two unread words, kept in registers

    const long long n_rows = a.n_sym == 0 ? 0 : (a.n_sym - 1) / a.index_div + 1;
    for (long long row = 0; row < n_rows; ++row) {
        const int32_t *cdf = a.cdfs + row * a.cdf_stride;
        // wave-uniform row constants, forced into scalar registers NOW: a vector load first used inside the symbol
        // loop would put an s_waitcnt vmcnt(0) there and drain the output stores on every chunk
        const int size = __builtin_amdgcn_readfirstlane(a.cdf_sizes[row]);
        const int max_value = size - 2;
        const int offset = __builtin_amdgcn_readfirstlane(a.offsets[row]);
        __syncthreads();  // previous row's table reads are done
        for (int k = lane; k + 1 < size; k += 64) {
            const uint32_t lo = (uint32_t)cdf[k], hi = (uint32_t)cdf[k + 1];
            rowtab[k] = (lo & 0xFFFFu) | ((hi - lo) << 16);
        }
        for (int k = 0; k + 1 < size; ++k) {
            const uint32_t lo = (uint32_t)cdf[k], hi = (uint32_t)cdf[k + 1];
            for (uint32_t c = lo + lane; c < hi && c < 65536u; c += 64) lut[c] = (LutT)k;
        }
        __syncthreads();
        const long long i_end = (row + 1) * a.index_div < a.n_sym ? (row + 1) * a.index_div : a.n_sym;
        long long i = row * a.index_div;

        // exact per-symbol path (escapes, ragged tails, redo of rolled-back chunks); works on x, rp and the ring
        auto renorm_slow = [&]() {
            if (x < kRansL) {
                x = (x << 32) | ring(rp);
                rp += 1;
            }
        };
        auto get_bits = [&]() {
            const int val = (int)(x & kMaxBypassVal);
            x >>= kBypassPrecision;
            renorm_slow();
            return val;
        };
        auto step_slow = [&](long long pos) {
            if (__any(lp - rp < 14)) top_up();
            const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
            const int sidx = lut[cum_freq];
            const uint32_t sf = rowtab[sidx];
            x = (unsigned long long)(sf >> 16) * (x >> kPrecision) + cum_freq - (sf & 0xFFFFu);
            renorm_slow();
            int value = sidx;
            if (value == max_value) value = decode_escape(get_bits, max_value, corrupt);
            wsb[pos * 64] = (uint32_t)(value + offset);
        };

        constexpr int U = 8;
        while (i < i_end) {
            if (i + U > i_end) {   // ragged tail of the row
                step_slow(i);
                ++i;
                wq = ring(rp);
                wq1 = ring(rp + 1);
                continue;
            }
            if (__any(lp - rp < 12)) top_up();
            const unsigned long long x0 = x;
            const int rp0 = rp;
            uint32_t esc = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
                const uint32_t sidx = lut[cum_freq];
                const uint32_t sf = rowtab[sidx];
                x = (unsigned long long)(sf >> 16) * (x >> kPrecision) + cum_freq - (sf & 0xFFFFu);
                const bool need = (x >> 31) == 0ull;
                x = need ? ((x << 32) | wq) : x;
                rp += need ? 1 : 0;
                wq = need ? wq1 : wq;
                wq1 = ring(rp + 1);
                esc |= (sidx == (uint32_t)max_value) ? 1u : 0u;
                wsb[(i + u) * 64] = (uint32_t)((int)sidx + offset);
            }
            if (__any(esc != 0)) {   // roll the chunk back and decode it exactly
                x = x0;
                rp = rp0;
#pragma unroll 1
                for (int u = 0; u < U; ++u) step_slow(i + u);
                wq = ring(rp);
                wq1 = ring(rp + 1);
            }
            i += U;
        }
    }
    if (active) a.status[s] = corrupt | (rp > n_words ? kStatusCorrupt : 0);
}

// Round 5: ONE LDS round trip per symbol instead of two.  The chain of the decoder above is cum_freq -> symbol (lut) -> {start,
// freq} (rowtab) -> state: two DEPENDENT LDS reads, ~80 cycles each on a wave that has its SIMD to itself.  An exact
// one-lookup table needs 32 bits per cumulative frequency (256 KB per row); but the CDF row is wave-uniform for index_div
// consecutive symbols, so a BUCKETED table per row can be staged in LDS: 8 192 buckets of eight cumulative frequencies, 8 bytes
// each = 64 KB, the footprint of the byte-sized lut.  A bucket names the symbol A that owns its first value and, if A ends
// inside the bucket, its successor B = A + 1 (symbols are consecutive, so B's start is A's end):
//     dword 0 = start_A | (end_A - 1) << 16         dword 1 = (end_B - 1) | A << 16 | multi << 24
// (ends are stored minus one: 65 536 fits 16 bits).  One ds_read_b64, then selects: the symbol is B iff cum_freq > end_A - 1.
// `multi` marks a bucket that B does not finish either (two boundaries within eight values: symbols of probability < 2^-13,
// the far tails of a trained table) -- such a symbol, like an escape, sends its 8-symbol chunk to the exact path, which finds
// the symbol by bisection over the row's starts.  The tables are built by a parallel pre-pass into the workspace
// (rans_build_dec_table_kernel: one thread per bucket) and copied row by row into LDS.  Rows of up to 257 entries (symbol
// index < 256).  Same symbols out, bit for bit (tests/test_gpu_kernels.py: every decoder test runs through this kernel).
constexpr int kBucketShift = 3, kBucketsPerRow = 65536 >> kBucketShift;      // 8 192 buckets x 8 B = 64 KB per row

__global__ __launch_bounds__(256) void rans_build_dec_table_kernel(const int32_t *__restrict__ cdfs, const int32_t *__restrict__ cdf_sizes,
                                                                   int n_rows, int cdf_stride, uint2 *__restrict__ tab) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_rows * kBucketsPerRow) return;
    const int r = (int)(t / kBucketsPerRow), b = (int)(t - (long long)r * kBucketsPerRow);
    const int32_t *cdf = cdfs + (long long)r * cdf_stride;
    const int size = cdf_sizes[r];                  // entries; symbols 0 .. size - 2, cdf[size - 1] = 65536
    const unsigned cf0 = (unsigned)b << kBucketShift;
    int lo = 0, hi = size - 1;                      // A = last k with cdf[k] <= cf0
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned)cdf[mid] <= cf0) lo = mid; else hi = mid;
    }
    const int A = lo;
    const unsigned start_a = (unsigned)cdf[A], end_a = (unsigned)cdf[A + 1];
    const bool has_b = A + 2 <= size - 1;
    const unsigned end_b = has_b ? (unsigned)cdf[A + 2] : end_a;
    const unsigned multi = end_b < cf0 + (1u << kBucketShift) && end_b < 65536u ? 1u : 0u;
    uint2 e;
    e.x = (start_a & 0xFFFFu) | ((end_a - 1u) << 16);
    e.y = ((end_b - 1u) & 0xFFFFu) | ((unsigned)A << 16) | (multi << 24);
    tab[t] = e;
}

__global__ __launch_bounds__(64) void rans_dec_lut8_kernel(const RansArgs a, const uint2 *__restrict__ gtab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *tab = reinterpret_cast<uint2 *>(smem);                                            // [8192]
    uint32_t *win = reinterpret_cast<uint32_t *>(smem + kBucketsPerRow * 8);                 // [kWin][64]
    uint32_t *rowtab = win + kWin * 64;                                                      // start | freq << 16 per symbol
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    const int s = blk * 64 + lane;
    const bool active = s < a.n_streams;
    const int sc = active ? s : a.n_streams - 1;
    __builtin_amdgcn_s_setprio(SC2_RANS_PRIO);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.buf + (long long)sc * a.stride + a.io_offset[sc]);
    const int n_words = a.io_nbytes[sc] / 4;
    uint32_t *wsb = a.ws + (long long)blk * a.n_sym * 64 + lane;
    uint32_t *const wl = win + lane;

    int corrupt = 0;
    int lp = 0, rp = 0;
    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
    uint32_t pre[16];
    auto prefetch = [&]() {   // pre[] <- words [lp, lp + 16) of the lane's stream (zeros past its end)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = lp + 4 * q;
            Words4 t;
            if (k + 3 < n_words) {
                t = *reinterpret_cast<const Words4 *>(w + k);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t.v[e] = k + e < n_words ? w[k + e] : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[4 * q + e] = t.v[e];
        }
    };
    prefetch();
    auto top_up = [&]() {
        const bool want = lp - rp <= kWin - 16;
        if (want) {
#pragma unroll
            for (int j = 0; j < 16; ++j) wl[((lp + j) & (kWin - 1)) * 64] = pre[j];
            lp += 16;
            prefetch();
        }
    };
    auto ring = [&](int k) { return wl[(k & (kWin - 1)) * 64]; };
    top_up();
    top_up();
    unsigned long long x = (unsigned long long)ring(0) | ((unsigned long long)ring(1) << 32);
    rp = 2;
    uint32_t wq = ring(rp), wq1 = ring(rp + 1);

    const long long n_rows = a.n_sym == 0 ? 0 : (a.n_sym - 1) / a.index_div + 1;
    for (long long row = 0; row < n_rows; ++row) {
        const int32_t *cdf = a.cdfs + row * a.cdf_stride;
        const int size = __builtin_amdgcn_readfirstlane(a.cdf_sizes[row]);
        const int max_value = size - 2;
        const int offset = __builtin_amdgcn_readfirstlane(a.offsets[row]);
        __syncthreads();  // previous row's table reads are done
        {   // this row's 64 KB of buckets: 64 lanes x 16 B per trip, eight trips in flight
            const uint4 *src = reinterpret_cast<const uint4 *>(gtab + row * kBucketsPerRow) + lane;
            uint4 *dst = reinterpret_cast<uint4 *>(tab) + lane;
#pragma unroll 1
            for (int it = 0; it < kBucketsPerRow * 8 / 16 / 64; it += 8) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(it + u) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) dst[(it + u) * 64] = v[u];
            }
        }
        for (int k = lane; k + 1 < size; k += 64) {
            const uint32_t lo = (uint32_t)cdf[k], hi = (uint32_t)cdf[k + 1];
            rowtab[k] = (lo & 0xFFFFu) | ((hi - lo) << 16);
        }
        __syncthreads();
        const long long i_end = (row + 1) * a.index_div < a.n_sym ? (row + 1) * a.index_div : a.n_sym;
        long long i = row * a.index_div;

        // exact per-symbol path (escapes, multi-boundary buckets, ragged tails, redo of rolled-back chunks)
        auto renorm_slow = [&]() {
            if (x < kRansL) {
                x = (x << 32) | ring(rp);
                rp += 1;
            }
        };
        auto get_bits = [&]() {
            const int val = (int)(x & kMaxBypassVal);
            x >>= kBypassPrecision;
            renorm_slow();
            return val;
        };
        auto step_slow = [&](long long pos) {
            if (__any(lp - rp < 14)) top_up();
            const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
            int lo = 0, hi = size - 1;        // last symbol whose start <= cum_freq (bisection over the row's starts)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((rowtab[mid] & 0xFFFFu) <= cum_freq && mid <= max_value) lo = mid; else hi = mid;
            }
            const int sidx = lo;
            const uint32_t sf = rowtab[sidx];
            x = (unsigned long long)(sf >> 16) * (x >> kPrecision) + cum_freq - (sf & 0xFFFFu);
            renorm_slow();
            int value = sidx;
            if (value == max_value) value = decode_escape(get_bits, max_value, corrupt);
            wsb[pos * 64] = (uint32_t)(value + offset);
        };

        constexpr int U = 8;
        while (i < i_end) {
            if (i + U > i_end) {   // ragged tail of the row
                step_slow(i);
                ++i;
                wq = ring(rp);
                wq1 = ring(rp + 1);
                continue;
            }
            if (__any(lp - rp < 12)) top_up();
            const unsigned long long x0 = x;
            const int rp0 = rp;
            uint32_t bad = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
                const uint2 e = tab[cum_freq >> kBucketShift];
                const unsigned ea = e.x >> 16;
                const bool sel = cum_freq > ea;
                const unsigned start = sel ? ea + 1u : (e.x & 0xFFFFu);
                const unsigned nm1 = sel ? (e.y & 0xFFFFu) : ea;
                const uint32_t sidx = ((e.y >> 16) & 0xFFu) + (sel ? 1u : 0u);
                x = (unsigned long long)(nm1 + 1u - start) * (x >> kPrecision) + cum_freq - start;
                const bool need = (x >> 31) == 0ull;
                x = need ? ((x << 32) | wq) : x;
                rp += need ? 1 : 0;
                wq = need ? wq1 : wq;
                wq1 = ring(rp + 1);
                bad |= (e.y >> 24) | ((sidx == (uint32_t)max_value) ? 1u : 0u);
                wsb[(i + u) * 64] = (uint32_t)((int)sidx + offset);
            }
            if (__any(bad != 0)) {   // roll the chunk back and decode it exactly
                x = x0;
                rp = rp0;
#pragma unroll 1
                for (int u = 0; u < U; ++u) step_slow(i + u);
                wq = ring(rp);
                wq1 = ring(rp + 1);
            }
            i += U;
        }
    }
    if (active) a.status[s] = corrupt | (rp > n_words ? kStatusCorrupt : 0);
}

// parallel pass: symbols_out[s][i] = ws[blk][i][lane]
__global__ __launch_bounds__(256) void rans_dec_finish_kernel(const RansArgs a) {
    __shared__ uint32_t tile[64][65];
    const int blk = blockIdx.y;
    const long long i0 = (long long)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int pl = t & 63;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int p = r * 4 + (t >> 6);
        const long long i = i0 + p;
        tile[p][pl] = i < a.n_sym ? a.ws[((long long)blk * a.n_sym + i) * 64 + pl] : 0u;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int sl = r * 4 + (t >> 6);
        const int s = blk * 64 + sl;
        const long long i = i0 + pl;
        if (s < a.n_streams && i < a.n_sym) a.symbols_out[(long long)s * a.n_sym + i] = (int32_t)tile[pl][sl];
    }
}

// parallel pass of the fused form: the decoded symbols leave as the DEQUANTISED latent the synthesis transform reads,
// y_hat[s][pix][c] = bf16(symbol + median[c]) (EntropyModel.dequantize of sc2bench/models/layer.py:520 + the bf16 NHWC copy of
// eb_dequantize_tile_kernel, same rounding), straight from the [position][lane] intermediate: the int32 symbols (8 x 74 MB per
// coder launch of the bench) are never written and the dequantise launch of the decoder + head stage disappears.
// Block = 64 streams x P pixels x all C channels; position i = c * HW + pix (implicit indexes, index_div = HW).
constexpr int kDqP = 16;
__global__ __launch_bounds__(256) void rans_dec_finish_dq_kernel(const RansArgs a, const float *__restrict__ medians,
                                                                 uint16_t *__restrict__ y_hat, int C, int HW) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dq[];   // [64][LS]: stream-major, (pixel, channel) bf16 inside
    const int LS = kDqP * C * 2 + 16;                                    // (C % 8 == 0: rows stay 16-byte aligned)
    const int blk = blockIdx.y;
    const int pix0 = blockIdx.x * kDqP;
    const int np = HW - pix0 < kDqP ? HW - pix0 : kDqP;
    const int t = threadIdx.x, pl = t & 63;
    for (int row = t >> 6; row < C * kDqP; row += 4) {
        const int c = row / kDqP, p = row - c * kDqP;
        if (p < np) {
            const long long i = (long long)c * HW + pix0 + p;
            const int32_t v = (int32_t)a.ws[((long long)blk * a.n_sym + i) * 64 + pl];
            *reinterpret_cast<uint16_t *>(dq + pl * LS + (p * C + c) * 2) = f32_to_bf16_bits((float)v + medians[c]);
        }
    }
    __syncthreads();
    const int chunks = np * C * 2 / 16;   // 16-byte chunks per stream, contiguous in the NHWC output
    for (int q = t; q < 64 * chunks; q += 256) {
        const int sl = q / chunks, k = q - sl * chunks;
        const int s = blk * 64 + sl;
        if (s < a.n_streams)
            *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(y_hat) + ((long long)s * HW + pix0) * C * 2 + k * 16) =
                *reinterpret_cast<const uint4 *>(dq + sl * LS + k * 16);
    }
}

// The same pass WITHOUT LDS (round 5), for channel counts of at most 32 (the FP bottleneck's 24): one wave = 64 streams x two pixels,
// lane = stream.  A lane reads its stream's 2 C symbols of the pixel pair (for a fixed position the 64 lanes' words are 256
// contiguous bytes of the intermediate: every load is coalesced, 2 C of them independent and in flight together) and writes them as
// one contiguous run of 2 C bf16 (96 bytes at C = 24) of the NHWC latent.  Why: the form above needs 50 KB of LDS per workgroup, and
// inside the pipelined bench the CUs' LDS is held by the persistent convolution kernels of the neighbouring steps (conv2x2_gdn512:
// 152 KB): its 6 080 workgroups per coder launch trickled in behind them -- 3.3 ms per launch of 8 batches for 0.9 GB of traffic
// (0.41 ms per step in `bottleneck_forward`), 0.03 ms per batch alone.  A kernel without LDS fits beside anything.
template <int C>
__global__ __launch_bounds__(256) void rans_dec_finish_dq_reg_kernel(const RansArgs a, const float *__restrict__ medians,
                                                                     uint16_t *__restrict__ y_hat, int HW) {
    static_assert(C % 8 == 0 && C <= 32, "8-channel groups, at most 32 channels");
    const int blk = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int pix0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (pix0 >= HW) return;
    const int s = blk * 64 + lane;
    const bool two = pix0 + 1 < HW;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.ws) + ((long long)blk * a.n_sym + pix0) * 64 + lane;   // + (c HW + p) * 64
    int32_t v[2][C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        v[0][c] = (int32_t)w[(long long)c * HW * 64];
        v[1][c] = two ? (int32_t)w[((long long)c * HW + 1) * 64] : 0;
    }
    if (s >= a.n_streams) return;
    uint4 *out = reinterpret_cast<uint4 *>(y_hat + ((long long)s * HW + pix0) * C);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p == 1 && !two) break;
#pragma unroll
        for (int g = 0; g < C / 8; ++g) {
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = g * 8 + 2 * e;
                o[e] = (uint32_t)f32_to_bf16_bits((float)v[p][c] + medians[c]) | ((uint32_t)f32_to_bf16_bits((float)v[p][c + 1] + medians[c + 1]) << 16);
            }
            out[p * (C / 8) + g] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// generic decoder: explicit per-symbol indexes (or rows too long for the LUT path); upper-bound binary search,
// identical in result to upstream's linear find_if over the strictly increasing CDF row.
template <bool LDS_TABLES>
__global__ __launch_bounds__(64) void rans_dec_generic_kernel(const RansArgs a) {
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
    DecState d;
    const int n_words = a.io_nbytes[s] / 4;
    int corrupt = 0;
    dec_init(d, w, n_words);
    for (long long i = 0; i < a.n_sym; ++i) {
        const int idx = idxp ? idxp[i] : (int)(i / a.index_div);
        const int32_t *cdf = tab + idx * a.cdf_stride;
        const int size = a.cdf_sizes[idx];
        const int max_value = size - 2;
        const unsigned cum_freq = (unsigned)(d.x & 0xFFFFu);
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
        if (value == max_value) value = dec_escape(d, max_value, corrupt);
        out[i] = value + a.offsets[idx];
    }
    a.status[s] = corrupt | (dec_past_end(d, w, n_words) ? kStatusCorrupt : 0);
}

// parallel pass: dst[blk][i][lane] = src[blk*64 + lane][i] (per-symbol indexes into the serial decoder's lane-contiguous
// order; `dst` = the workspace, n_blocks * n_sym * 64 words)
__global__ __launch_bounds__(256) void rans_transpose_in_kernel(const int32_t *__restrict__ src, int n_streams,
                                                                long long n_sym, uint32_t *__restrict__ dst) {
    __shared__ uint32_t tile[64][65];
    const int blk = blockIdx.y;
    const long long i0 = (long long)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int pl = t & 63;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int sl = r * 4 + (t >> 6);
        const int s = blk * 64 + sl;
        const long long i = i0 + pl;
        tile[pl][sl] = (s < n_streams && i < n_sym) ? (uint32_t)src[(long long)s * n_sym + i] : 0u;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int p = r * 4 + (t >> 6);
        const long long i = i0 + p;
        if (i < n_sym) dst[((long long)blk * n_sym + i) * 64 + pl] = tile[p][pl];
    }
}

// Decoder for explicit per-symbol indexes over a RAGGED table (the Gaussian conditional model: 64 rows of 5 .. 3133
// entries, 27 k entries in all although the rectangular table has 200 k): the rows are packed back to back into LDS as
// u16 (the closing 65536 of a row is implied by its position), so the upper-bound search - identical in result to
// upstream's linear find_if - runs on LDS instead of global memory.  Falls back to the global table inside the same
// kernel when the packed rows do not fit.  Indexes are fetched eight symbols ahead (they do not depend on the state).
constexpr int kRaggedCap = 61440;    // u16 entries of packed CDF rows kept in LDS (120 KB)
constexpr int kRaggedRows = 256;

template <int kBucketBits>
__global__ __launch_bounds__(64) void rans_dec_ragged_kernel(const RansArgs a) {
    constexpr int kBuckets = 1 << kBucketBits;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *l_cdf = reinterpret_cast<uint16_t *>(smem);                       // [kRaggedCap]
    int *row_start = reinterpret_cast<int *>(smem + kRaggedCap * 2);            // [n_cdfs + 1]
    int *l_size = row_start + a.n_cdfs + 1;                                     // [n_cdfs]
    int *l_off = l_size + a.n_cdfs;                                             // [n_cdfs]
    // bucket[row][b] = first k with cdf[k] >= b << (16 - bits): the search for cum_freq starts inside the bucket
    // of its top `bits` bits - a Gaussian row has most of its entries in the tails, most of its mass in a few buckets
    uint16_t *bucket = reinterpret_cast<uint16_t *>(l_off + a.n_cdfs);          // [n_cdfs][kBuckets + 1]
    const int lane = threadIdx.x;
    for (int r = lane; r < a.n_cdfs; r += 64) {
        l_size[r] = a.cdf_sizes[r];
        l_off[r] = a.offsets[r];
    }
    __syncthreads();
    if (lane == 0) {
        int tot = 0;
        for (int r = 0; r < a.n_cdfs; ++r) {
            row_start[r] = tot;
            tot += l_size[r];
        }
        row_start[a.n_cdfs] = tot;
    }
    __syncthreads();
    const bool in_lds = row_start[a.n_cdfs] <= kRaggedCap;
    if (in_lds) {
        for (int r = 0; r < a.n_cdfs; ++r) {
            const int size = l_size[r], st = row_start[r];
            for (int k = lane; k < size; k += 64) l_cdf[st + k] = (uint16_t)a.cdfs[(long long)r * a.cdf_stride + k];
        }
    }
    __syncthreads();
    if (in_lds) {
        for (int t = lane; t < a.n_cdfs * (kBuckets + 1); t += 64) {
            const int r = t / (kBuckets + 1), b = t - r * (kBuckets + 1);
            const int size = l_size[r];
            const uint16_t *row = l_cdf + row_start[r];
            const unsigned target = (unsigned)b << (16 - kBucketBits);   // first k with cdf[k] > target - 1, i.e. >= target
            int lo = 0, hi = size;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const unsigned v = mid == size - 1 ? 65536u : (unsigned)row[mid];
                if (v >= target && !(b == 0 && v == 0u)) hi = mid; else lo = mid + 1;
            }
            bucket[t] = (uint16_t)lo;
        }
    }
    __syncthreads();
    const int s = blockIdx.x * 64 + lane;
    if (s >= a.n_streams) return;
    __builtin_amdgcn_s_setprio(SC2_RANS_PRIO);
    // indexes arrive transposed to [position][lane] in the workspace; each slot is overwritten IN PLACE by the decoded
    // value of its position (an index is fetched >= 8 positions before its slot is written), and the finish pass
    // transposes the workspace into symbols_out
    uint32_t *out = a.ws + (long long)blockIdx.x * a.n_sym * 64 + lane;
    const uint32_t *idxp = out;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.buf + (long long)s * a.stride + a.io_offset[s]);
    DecState d;
    const int n_words = a.io_nbytes[s] / 4;
    int corrupt = 0;
    dec_init(d, w, n_words);
    constexpr int U = 8;
    int nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nxt[u] = u < a.n_sym ? (int)idxp[(long long)u * 64] : 0;
    for (long long i0 = 0; i0 < a.n_sym; i0 += U) {
        int cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cur[u] = nxt[u];
            nxt[u] = i0 + U + u < a.n_sym ? (int)idxp[(i0 + U + u) * 64] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < a.n_sym) {
                const int idx = cur[u];
                const int size = l_size[idx];
                const int max_value = size - 2;
                const unsigned cum_freq = (unsigned)(d.x & 0xFFFFu);
                int lo = 0, hi = size;
                unsigned start, next;
                if (in_lds) {
                    const uint16_t *row = l_cdf + row_start[idx];
                    const uint16_t *bk = bucket + idx * (kBuckets + 1) + (cum_freq >> (16 - kBucketBits));
                    lo = bk[0];                                   // first k with cdf[k] >= bucket floor (<= answer)
                    hi = min((int)bk[1] + 1, size);               // the answer is <= first k with cdf[k] >= next floor
                    while (lo < hi) {   // first k with cdf[k] > cum_freq; the last entry of a row is 65536
                        const int mid = (lo + hi) >> 1;
                        const unsigned v = mid == size - 1 ? 65536u : (unsigned)row[mid];
                        if (v > cum_freq) hi = mid; else lo = mid + 1;
                    }
                    start = (unsigned)row[lo - 1];
                    next = lo == size - 1 ? 65536u : (unsigned)row[lo];
                } else {
                    const int32_t *row = a.cdfs + (long long)idx * a.cdf_stride;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if ((unsigned)row[mid] > cum_freq) hi = mid; else lo = mid + 1;
                    }
                    start = (unsigned)row[lo - 1];
                    next = (unsigned)row[lo];
                }
                const int sidx = lo - 1;
                d.x = (unsigned long long)(next - start) * (d.x >> kPrecision) + cum_freq - start;
                dec_renorm(d);
                int value = sidx;
                if (value == max_value) value = dec_escape(d, max_value, corrupt);
                out[(i0 + u) * 64] = (uint32_t)(value + l_off[idx]);
            }
        }
    }
    a.status[s] = corrupt | (dec_past_end(d, w, n_words) ? kStatusCorrupt : 0);
}

// Round 4: the same decoder (explicit per-symbol indexes over a ragged table, rows packed into LDS as u16), rebuilt around what
// bounds a serial wave: INSTRUCTION ISSUE.  One lane per stream spent ~100 vector instructions per symbol on a search whose
// probes are independent -- so here FOUR lanes serve a stream (a quad; 16 streams per wave), every lane keeps a copy of the state
// and the quad splits the search:
//   * ONE 2-byte bucket read: first candidate entry of the 256-wide cum_freq bucket;
//   * lane q probes candidates q + 1 and q + 5 (eight candidates per symbol, two u16 reads per lane), the count of candidates
//     <= cum_freq is a quad sum (two DPP adds);
//   * {start, next} = ONE unaligned 4-byte read at the answer (entries lo + cnt - 1, lo + cnt);
//   * stream words come from an LDS ring per stream, topped up 16 words at a time one batch ahead, every lane of the quad moving
//     four of them (no global load and no pointer compare on the state's chain);
//   * 8-symbol chunks are decoded speculatively and branch-free, and rolled back to the exact per-symbol path when any stream of
//     the wave met an escape symbol or a bucket with more than eight candidates (EVERY row crowds its tail entries into the
//     first and the last bucket and cum_freq is uniform: with four candidates ~0.8 % of the symbols were unresolved and
//     practically every chunk rolled back).
// Rows are stored as cdf[k] - 1 (the closing 65536 fits; "cdf > cum_freq" is "stored >= cum_freq") and each is followed by eight
// 0xFFFF sentinels: probes need no clamping.  Per-symbol row metadata is read ahead of the chain with the indexes.
// y decode of the MSHP bottleneck, 256 x 72 600 symbols: 43.5 ms (one lane per stream, global stream words) -> 26.7 (ring +
// speculation, one lane per stream, ~97 instructions per symbol) -> 19.4 ms (quads: ~58 instructions and three dependent LDS
// round trips per symbol -- bucket, probes, {start, next} -- which is now what a symbol costs: ~570 cycles).  One wave per
// workgroup (each with its own copy of the tables) is the fastest arrangement measured: 19.4 / 19.5 / 20.9 / 26.0 ms at 1 / 2 /
// 4 / 8 waves per workgroup -- the chip is otherwise idle, and waves that share a CU share its LDS pipe.
// Same bytes in, same symbols out (tests/test_gpu_hyperprior.py, tests/test_gpu_kernels.py).
constexpr int kRagged2Cap = 45056;   // u16 entries of packed CDF rows incl. sentinels (88 KB) beside 32 KB of buckets and the 8 KB ring
constexpr int kRagged2Rows = 64;
constexpr int kProbe = 8;            // candidate entries examined by the speculative path

__device__ __forceinline__ int quad_sum(int v) {   // sum over the four lanes of a quad, in every lane
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
    return v;
}

template <int WAVES>   // waves (of 16 streams) per workgroup, which share one copy of the tables
__global__ __launch_bounds__(64 * WAVES) void rans_dec_ragged2_kernel(const RansArgs a) {
    constexpr int T = 64 * WAVES, SPW = 16 * WAVES;
    constexpr int kBuckets = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *l_cdf = reinterpret_cast<uint16_t *>(smem);                                   // [kRagged2Cap]: cdf[k] - 1 (mod 2^16)
    uint16_t *bucket = reinterpret_cast<uint16_t *>(smem + kRagged2Cap * 2);                // [n_cdfs][257]: first candidate of a bucket
    uint32_t *win = reinterpret_cast<uint32_t *>(smem + kRagged2Cap * 2 + ((kRagged2Rows * (kBuckets + 1) * 2 + 15) & ~15));   // [kWin][SPW]
    int *row_start = reinterpret_cast<int *>(win + kWin * SPW);                              // [n_cdfs + 1]
    int *l_size = row_start + kRagged2Rows + 1;                                             // [n_cdfs]
    int *l_off = l_size + kRagged2Rows;                                                     // [n_cdfs]
    const int tid = threadIdx.x;
    for (int r = tid; r < a.n_cdfs; r += T) {
        l_size[r] = a.cdf_sizes[r];
        l_off[r] = a.offsets[r];
    }
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int r = 0; r < a.n_cdfs; ++r) {
            row_start[r] = tot;
            tot += l_size[r] + kProbe;
        }
        row_start[a.n_cdfs] = tot;
    }
    __syncthreads();
    const bool in_lds = row_start[a.n_cdfs] <= kRagged2Cap;   // (otherwise every symbol takes the exact path on the global table)
    if (in_lds) {
        for (int r = 0; r < a.n_cdfs; ++r) {
            const int size = l_size[r], st = row_start[r];
            for (int k = tid; k < size + kProbe; k += T)
                l_cdf[st + k] = k < size ? (uint16_t)((uint32_t)a.cdfs[(long long)r * a.cdf_stride + k] - 1u) : (uint16_t)0xFFFFu;
        }
    }
    __syncthreads();
    auto row_at = [&](const uint16_t *row, int k) -> unsigned { return k == 0 ? 0u : (unsigned)row[k] + 1u; };   // cdf[k]
    if (in_lds) {
        // bucket[r][b] = first k >= 1 with cdf[k] >= b << 8 (entries before it are < the bucket's floor <= cum_freq); bucket 256 = size - 1
        for (int t = tid; t < a.n_cdfs * (kBuckets + 1); t += T) {
            const int r = t / (kBuckets + 1), b = t - r * (kBuckets + 1);
            const int size = l_size[r];
            const uint16_t *row = l_cdf + row_start[r];
            const unsigned target = (unsigned)b << 8;
            int lo = 1, hi = size - 1;   // cdf[size - 1] = 65536 >= every target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (row_at(row, mid) >= target) hi = mid; else lo = mid + 1;
            }
            bucket[t] = (uint16_t)lo;
        }
    }
    __syncthreads();
    const int sl = tid >> 2, q = tid & 3;        // stream of the workgroup, lane of the quad
    const int s = blockIdx.x * SPW + sl;
    const int blk = s >> 6;                      // 64-stream block of the transposed workspace (a wave's 16 streams share it)
    if (blk * 64 >= a.n_streams) return;         // (a whole wave, past the last block; no workgroup barrier follows)
    const bool active = s < a.n_streams;
    const int sc = active ? s : a.n_streams - 1;
    __builtin_amdgcn_s_setprio(SC2_RANS_PRIO);
    uint32_t *out = a.ws + (long long)blk * a.n_sym * 64 + (s & 63);   // (the workspace has all 64 columns of every block)
    const uint32_t *idxp = out;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a.buf + (long long)sc * a.stride + a.io_offset[sc]);
    // a column without a stream decodes an all-zero stream with index 0 (x stays 0: entry 0 of row 0 for ever): it takes part in
    // the wave's votes without ever asking for the slow path
    const int n_words = active ? a.io_nbytes[sc] / 4 : 0;
    uint32_t *const wl = win + sl;

    // ---- the stream-word ring of rans_dec_lut_kernel, one per stream; lane q of the quad moves words 4 q .. 4 q + 3 of a batch
    int lp = 0, rp = 0, corrupt = 0;
    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
    uint32_t pre[4];
    auto prefetch = [&]() {   // pre[] <- words [lp + 4 q, lp + 4 q + 4)
        const int k = lp + 4 * q;
        Words4 t;
        if (k + 3 < n_words) {
            t = *reinterpret_cast<const Words4 *>(w + k);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) t.v[e] = k + e < n_words ? w[k + e] : 0u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[e] = t.v[e];
    };
    prefetch();
    auto top_up = [&]() {   // (lp, rp are equal in the four lanes of a quad: the quad tops up or not as one)
        if (lp - rp <= kWin - 16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wl[((lp + 4 * q + j) & (kWin - 1)) * SPW] = pre[j];
            lp += 16;
            prefetch();
        }
    };
    auto ring = [&](int k) { return wl[(k & (kWin - 1)) * SPW]; };
    top_up();
    top_up();
    unsigned long long x = (unsigned long long)ring(0) | ((unsigned long long)ring(1) << 32);
    rp = 2;
    uint32_t wq = ring(rp), wq1 = ring(rp + 1);

    // ---- exact per-symbol path (every lane of the quad runs it on its copy of the state): escapes, buckets with more than
    //      kProbe candidates, the ragged tail
    auto renorm_slow = [&]() {
        if (x < kRansL) {
            x = (x << 32) | ring(rp);
            rp += 1;
        }
    };
    auto get_bits = [&]() {
        const int val = (int)(x & kMaxBypassVal);
        x >>= kBypassPrecision;
        renorm_slow();
        return val;
    };
    auto step_slow = [&](long long pos, int idx) {
        if (__any(lp - rp < 14)) top_up();
        const int size = l_size[idx];
        const int max_value = size - 2;
        const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
        unsigned start, next;
        int lo, hi;
        if (in_lds) {
            const uint16_t *row = l_cdf + row_start[idx];
            const uint16_t *bk = bucket + idx * (kBuckets + 1) + (cum_freq >> 8);
            lo = (int)bk[0];
            hi = (int)bk[1];            // cdf[hi] >= the next bucket's floor > cum_freq: the answer is in [lo, hi]
            while (lo < hi) {           // first k with cdf[k] > cum_freq
                const int mid = (lo + hi) >> 1;
                if (row_at(row, mid) > cum_freq) hi = mid; else lo = mid + 1;
            }
            start = row_at(row, lo - 1);
            next = row_at(row, lo);
        } else {
            const int32_t *row = a.cdfs + (long long)idx * a.cdf_stride;
            lo = 0, hi = size;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((unsigned)row[mid] > cum_freq) hi = mid; else lo = mid + 1;
            }
            start = (unsigned)row[lo - 1];
            next = (unsigned)row[lo];
        }
        x = (unsigned long long)(next - start) * (x >> kPrecision) + cum_freq - start;
        renorm_slow();
        int value = lo - 1;
        if (value == max_value) value = decode_escape(get_bits, max_value, corrupt);
        out[pos * 64] = (uint32_t)(value + l_off[idx]);   // (the four lanes store the same word)
    };

    constexpr int U = 8;
    int nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nxt[u] = u < a.n_sym ? (int)idxp[(long long)u * 64] : 0;
    for (long long i0 = 0; i0 < a.n_sym; i0 += U) {
        int cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cur[u] = nxt[u];
            nxt[u] = i0 + U + u < a.n_sym ? (int)idxp[(i0 + U + u) * 64] : 0;
        }
        if (i0 + U > a.n_sym || !in_lds) {   // ragged tail (or tables that do not fit): exact path
#pragma unroll 1
            for (int u = 0; u < U; ++u)
                if (i0 + u < a.n_sym) step_slow(i0 + u, cur[u]);
            wq = ring(rp);
            wq1 = ring(rp + 1);
            continue;
        }
        if (__any(lp - rp < 12)) top_up();
        // row metadata of the chunk, off the state's chain
        int r_st[U], r_size[U], r_off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r_st[u] = row_start[cur[u]];
            r_size[u] = l_size[cur[u]];
            r_off[u] = l_off[cur[u]];
        }
        const unsigned long long x0 = x;
        const int rp0 = rp;
        uint32_t bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned cum_freq = (unsigned)(x & 0xFFFFu);
            const int lo = (int)bucket[cur[u] * (kBuckets + 1) + (int)(cum_freq >> 8)];
            const uint16_t *pr = l_cdf + r_st[u] + lo - 1;       // stored entries lo - 1 .. lo + 8 (sentinels behind the row)
            // candidate t (entry lo - 1 + t, t = 1 .. 8) is <= cum_freq  <=>  stored < cum_freq; entries increase, so those form a
            // prefix and their count is the answer's distance from lo.  This lane: candidates q + 1 and q + 5
            const int local = ((unsigned)pr[q + 1] < cum_freq ? 1 : 0) + ((unsigned)pr[q + 5] < cum_freq ? 1 : 0);
            const int cnt = quad_sum(local);
            uint32_t pair;                                       // stored[lo - 1 + cnt] | stored[lo + cnt] << 16
            __builtin_memcpy(&pair, pr + cnt, 4);
            const unsigned start = (cnt == 0 && lo == 1) ? 0u : (pair & 0xFFFFu) + 1u;   // entry 0 of a row is cdf 0 (stored 0xFFFF)
            const unsigned next = (pair >> 16) + 1u;
            const int sidx = lo + cnt - 1;
            bad |= (active && (cnt == kProbe || sidx == r_size[u] - 2)) ? 1u : 0u;
            x = (unsigned long long)(next - start) * (x >> kPrecision) + cum_freq - start;
            const bool need = (x >> 31) == 0ull;
            x = need ? ((x << 32) | wq) : x;
            rp += need ? 1 : 0;
            wq = need ? wq1 : wq;
            wq1 = ring(rp + 1);
            out[(i0 + u) * 64] = (uint32_t)(sidx + r_off[u]);
        }
        if (__any(bad != 0)) {   // roll the chunk back and decode it exactly
            x = x0;
            rp = rp0;
#pragma unroll 1
            for (int u = 0; u < U; ++u) step_slow(i0 + u, cur[u]);
            wq = ring(rp);
            wq1 = ring(rp + 1);
        }
    }
    if (active && q == 0) a.status[s] = corrupt | (rp > n_words ? kStatusCorrupt : 0);
}

int check_common(const int32_t *indexes, long long index_div, int n_streams, long long n_sym, const int32_t *cdfs,
                 int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets) {
    SC2_REQUIRE(cdfs && cdf_sizes && offsets, SC2_ERR_INVALID_ARG, "rans: Uninitialized CDFs. Run update() first");
    SC2_REQUIRE(n_streams > 0 && n_sym >= 0 && n_cdfs > 0 && cdf_stride >= 3, SC2_ERR_INVALID_ARG,
                "rans: bad sizes n_streams=%d n_sym=%lld n_cdfs=%d cdf_stride=%d", n_streams, n_sym, n_cdfs,
                cdf_stride);
    SC2_REQUIRE((long long)n_cdfs * cdf_stride < (1ll << 31), SC2_ERR_UNSUPPORTED, "rans: CDF table too large");
    if (!indexes) SC2_REQUIRE(index_div > 0, SC2_ERR_INVALID_ARG, "rans: index_div must be positive");
    if (!indexes && n_sym > 0)
        SC2_REQUIRE((n_sym - 1) / index_div < n_cdfs, SC2_ERR_INVALID_ARG,
                    "rans: implicit index %lld out of range (%d CDF rows)", (n_sym - 1) / index_div, n_cdfs);
    return SC2_OK;
}

// Small launches (<= 16 serial waves: up to 1 024 streams) ask for 159 KB of LDS per wave, so that no LDS-using workgroup of
// another kernel shares their CU: the serial chain is latency-bound, every MFMA wave on its SIMD stretches it, and a launch this
// small is what the FIRST decoder stage of a pipeline run waits for (bench K = 20: first coder launch 28.0 -> 27.3 ms, + 0.7 %
// images/s, 3 of 3 runs; 16 CUs at most).  Larger launches keep their footprint: giving 32 - 64 CUs away costs more than it
// gains.  SC2_RANS_LDS_PAD (KB, 0 = off) / SC2_RANS_PAD_WAVES override (A/B).
static size_t lds_pad(size_t need, int n_wave_blocks) {
    const int pad_kb = sc2_pol().rans_lds_pad_kb, max_waves = sc2_pol().rans_pad_waves;
    const size_t want = (size_t)pad_kb * 1024;
    return (n_wave_blocks <= max_waves && want > need) ? want : need;
}

template <class K>
void allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

// the dequantising last pass of a decode launch: the register form where it applies, else the LDS form
int launch_finish_dq(const RansArgs &a, const float *medians, void *y_hat, int n_cdfs, int HW, int n_blocks, hipStream_t s,
                            void *ev_dq_begin, void *ev_dq_end) {
    if (ev_dq_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_dq_begin), s);
    const bool reg = (n_cdfs == 24 || n_cdfs == 16 || n_cdfs == 8 || n_cdfs == 32) && !sc2_pol().rans_dq_lds;   // (rans_dq_lds = 1: A/B)
    if (reg) {
        const dim3 grid((unsigned)((HW + 7) / 8), (unsigned)n_blocks);
        uint16_t *y = static_cast<uint16_t *>(y_hat);
        switch (n_cdfs) {
            case 8: hipLaunchKernelGGL(rans_dec_finish_dq_reg_kernel<8>, grid, dim3(256), 0, s, a, medians, y, HW); break;
            case 16: hipLaunchKernelGGL(rans_dec_finish_dq_reg_kernel<16>, grid, dim3(256), 0, s, a, medians, y, HW); break;
            case 24: hipLaunchKernelGGL(rans_dec_finish_dq_reg_kernel<24>, grid, dim3(256), 0, s, a, medians, y, HW); break;
            default: hipLaunchKernelGGL(rans_dec_finish_dq_reg_kernel<32>, grid, dim3(256), 0, s, a, medians, y, HW); break;
        }
    } else {
        const size_t dq_lds = (size_t)64 * (kDqP * n_cdfs * 2 + 16);
        allow_big_lds(rans_dec_finish_dq_kernel, dq_lds);
        hipLaunchKernelGGL(rans_dec_finish_dq_kernel, dim3((HW + kDqP - 1) / kDqP, n_blocks), dim3(256), dq_lds, s, a, medians,
                           static_cast<uint16_t *>(y_hat), n_cdfs, HW);
    }
    if (ev_dq_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_dq_end), s);
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}


}  // namespace

extern "C" int64_t sc2_rans_max_bytes(int64_t n_sym) {
    // worst case per symbol: 16 bits (freq 1) + escape: count nibble + 8 raw nibbles = 52 bits; + 8 flush bytes.
    if (n_sym < 0) return 0;
    const int64_t words = (n_sym * 52 + 31) / 32 + 4;
    return words * 4;
}

static int64_t ws_entries_bytes(int n_streams, int64_t n_sym) {
    const int64_t n_blocks = (n_streams + 63) / 64;
    const int64_t b = n_blocks * n_sym * 64 * 4;
    return (b + 255) / 256 * 256 + 256;
}

// the bucketed one-lookup decode tables (rans_dec_lut8_kernel): 64 KB per CDF row, rows of up to 257 entries, at most 1 024 rows
static int64_t dec_table_bytes(int n_cdfs, int cdf_stride) {
    return (cdf_stride <= 257 && n_cdfs <= 1024) ? (int64_t)n_cdfs * kBucketsPerRow * 8 : 0;
}

extern "C" int64_t sc2_rans_workspace_bytes(int n_streams, int64_t n_sym, int n_cdfs, int cdf_stride) {
    if (n_streams <= 0 || n_sym < 0 || n_cdfs <= 0 || cdf_stride <= 0) return 0;
    const int64_t tables = (int64_t)n_cdfs * cdf_stride * (int64_t)sizeof(EncEntry);
    const int64_t dec = dec_table_bytes(n_cdfs, cdf_stride);
    return ws_entries_bytes(n_streams, n_sym) + (tables > dec ? tables : dec);
}

extern "C" int sc2_rans_encode_batch(const int32_t *symbols, const int32_t *indexes, int64_t index_div, int n_streams,
                                     int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                     const int32_t *cdf_sizes, const int32_t *offsets, uint8_t *out,
                                     int64_t out_stride, int32_t *out_offset, int32_t *out_nbytes, int32_t *status,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    int rc = check_common(indexes, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets);
    if (rc != SC2_OK) return rc;
    SC2_REQUIRE((symbols || n_sym == 0) && out && out_offset && out_nbytes && status && workspace,
                SC2_ERR_INVALID_ARG, "rans_encode: null argument");
    SC2_REQUIRE(out_stride >= 16 && out_stride % 4 == 0 && out_stride < (1ll << 31), SC2_ERR_INVALID_ARG,
                "rans_encode: out_stride %lld must be a multiple of 4 in [16, 2^31)", (long long)out_stride);
    SC2_REQUIRE(workspace_bytes >= sc2_rans_workspace_bytes(n_streams, n_sym, n_cdfs, cdf_stride), SC2_ERR_INVALID_ARG,
                "rans_encode: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)sc2_rans_workspace_bytes(n_streams, n_sym, n_cdfs, cdf_stride));
    RansArgs a;
    a.symbols = symbols; a.indexes = indexes; a.index_div = index_div > 0 ? index_div : 1;
    a.n_streams = n_streams; a.n_sym = n_sym;
    a.cdfs = cdfs; a.n_cdfs = n_cdfs; a.cdf_stride = cdf_stride; a.cdf_sizes = cdf_sizes; a.offsets = offsets;
    a.buf = out; a.stride = out_stride; a.io_offset = out_offset; a.io_nbytes = out_nbytes; a.status = status;
    a.symbols_out = nullptr; a.ws = static_cast<uint32_t *>(workspace);
    const int n_blocks = (n_streams + 63) / 64;
    const int n_entries = n_cdfs * cdf_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_sym > 0) {
        const long long gx = (n_sym + 63) / 64;
        SC2_REQUIRE(gx < (1ll << 31) && n_blocks <= 65535, SC2_ERR_UNSUPPORTED, "rans_encode: problem too large");
        hipLaunchKernelGGL(rans_enc_prepare_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, a);
        SC2_CHECK_LAUNCH();
    }
    EncEntry *gtab = reinterpret_cast<EncEntry *>(static_cast<unsigned char *>(workspace) +
                                                  ws_entries_bytes(n_streams, n_sym));
    hipLaunchKernelGGL(rans_build_enc_table_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, s, cdfs, n_entries,
                       cdf_stride, gtab);
    SC2_CHECK_LAUNCH();
    const size_t stage_lds = (size_t)(kStage + 1) * 64 * 4;
    if (n_entries <= kEncLdsEntries) {
        const size_t lds = lds_pad(stage_lds + (size_t)n_entries * sizeof(EncEntry), n_blocks);
        allow_big_lds(rans_enc_serial_kernel<true>, lds);
        hipLaunchKernelGGL(rans_enc_serial_kernel<true>, dim3(n_blocks), dim3(64), lds, s, a, gtab);
    } else {
        hipLaunchKernelGGL(rans_enc_serial_kernel<false>, dim3(n_blocks), dim3(64), stage_lds, s, a, gtab);
    }
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

// medians / y_hat: the fused dequantised output of sc2_rans_decode_dequantize_batch (symbols_out may then be null)
static int decode_impl(const uint8_t *in, int64_t in_stride, const int32_t *in_offset,
                       const int32_t *in_nbytes, const int32_t *indexes, int64_t index_div,
                       int n_streams, int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                       const int32_t *cdf_sizes, const int32_t *offsets, int32_t *symbols_out,
                       int32_t *status, void *workspace, int64_t workspace_bytes, void *stream, const float *medians,
                       void *y_hat, void *ev_dq_begin = nullptr, void *ev_dq_end = nullptr) {
    int rc = check_common(indexes, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets);
    if (rc != SC2_OK) return rc;
    SC2_REQUIRE(in && in_offset && in_nbytes && (symbols_out || y_hat || n_sym == 0) && status && workspace,
                SC2_ERR_INVALID_ARG, "rans_decode: null argument");
    if (y_hat) {
        SC2_REQUIRE(medians, SC2_ERR_INVALID_ARG, "rans_decode_dequantize: null medians");
        SC2_REQUIRE(!indexes && cdf_stride <= kMaxRowLds && n_cdfs % 8 == 0 && n_cdfs <= 64 && n_sym == (int64_t)n_cdfs * index_div &&
                        index_div < (1ll << 31),
                    SC2_ERR_UNSUPPORTED,
                    "rans_decode_dequantize: needs implicit indexes, n_sym == n_cdfs * index_div and a channel count that is a "
                    "multiple of 8, <= 64 (got %d rows, %lld symbols, index_div %lld)", n_cdfs, (long long)n_sym, (long long)index_div);
    }
    SC2_REQUIRE(in_stride >= 8 && in_stride % 4 == 0, SC2_ERR_INVALID_ARG,
                "rans_decode: in_stride %lld must be a multiple of 4, >= 8", (long long)in_stride);
    SC2_REQUIRE(workspace_bytes >= sc2_rans_workspace_bytes(n_streams, n_sym, n_cdfs, cdf_stride), SC2_ERR_INVALID_ARG,
                "rans_decode: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)sc2_rans_workspace_bytes(n_streams, n_sym, n_cdfs, cdf_stride));
    RansArgs a;
    a.symbols = nullptr; a.indexes = indexes; a.index_div = index_div > 0 ? index_div : 1;
    a.n_streams = n_streams; a.n_sym = n_sym;
    a.cdfs = cdfs; a.n_cdfs = n_cdfs; a.cdf_stride = cdf_stride; a.cdf_sizes = cdf_sizes; a.offsets = offsets;
    a.buf = const_cast<uint8_t *>(in); a.stride = in_stride;
    a.io_offset = const_cast<int32_t *>(in_offset); a.io_nbytes = const_cast<int32_t *>(in_nbytes);
    a.status = status; a.symbols_out = symbols_out; a.ws = static_cast<uint32_t *>(workspace);
    const int n_blocks = (n_streams + 63) / 64;
    const int n_entries = n_cdfs * cdf_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lut8 = sc2_pol().rans_lut8 != 0;   // (0: A/B, the two-lookup decoder)
    if (!indexes && lut8 && dec_table_bytes(n_cdfs, cdf_stride) > 0) {
        // one LDS round trip per symbol: bucketed tables built by a parallel pre-pass into the workspace (behind the [position][lane]
        // intermediate), then the serial kernel
        const long long n_rows = n_sym == 0 ? 0 : (n_sym - 1) / a.index_div + 1;
        uint2 *gtab = reinterpret_cast<uint2 *>(static_cast<unsigned char *>(workspace) + ws_entries_bytes(n_streams, n_sym));
        if (n_rows > 0) {
            const long long threads = n_rows * kBucketsPerRow;
            hipLaunchKernelGGL(rans_build_dec_table_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, cdfs, cdf_sizes,
                               (int)n_rows, cdf_stride, gtab);
            SC2_CHECK_LAUNCH();
        }
        const size_t lds = lds_pad((size_t)kBucketsPerRow * 8 + (size_t)kWin * 64 * 4 + (size_t)cdf_stride * 4 + 16, n_blocks);
        allow_big_lds(rans_dec_lut8_kernel, lds);
        hipLaunchKernelGGL(rans_dec_lut8_kernel, dim3(n_blocks), dim3(64), lds, s, a, gtab);
        SC2_CHECK_LAUNCH();
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            SC2_REQUIRE(gx < (1ll << 31) && n_blocks <= 65535, SC2_ERR_UNSUPPORTED, "rans_decode: problem too large");
            if (y_hat) {
                if (const int rc = launch_finish_dq(a, medians, y_hat, n_cdfs, (int)index_div, n_blocks, s, ev_dq_begin, ev_dq_end)) return rc;
            }
            if (symbols_out) {
                hipLaunchKernelGGL(rans_dec_finish_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, a);
                SC2_CHECK_LAUNCH();
            }
        }
        return SC2_OK;
    }
    if (!indexes && cdf_stride <= kMaxRowLds) {
        const bool small = cdf_stride <= 257;   // symbol indexes fit a byte
        const size_t lds = lds_pad((size_t)65536 * (small ? 1 : 2) + (size_t)kWin * 64 * 4 + (size_t)cdf_stride * 4 + 16, n_blocks);
        if (small) {
            allow_big_lds(rans_dec_lut_kernel<uint8_t>, lds);
            hipLaunchKernelGGL(rans_dec_lut_kernel<uint8_t>, dim3(n_blocks), dim3(64), lds, s, a);
        } else {
            allow_big_lds(rans_dec_lut_kernel<uint16_t>, lds);
            hipLaunchKernelGGL(rans_dec_lut_kernel<uint16_t>, dim3(n_blocks), dim3(64), lds, s, a);
        }
        SC2_CHECK_LAUNCH();
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            SC2_REQUIRE(gx < (1ll << 31) && n_blocks <= 65535, SC2_ERR_UNSUPPORTED, "rans_decode: problem too large");
            if (y_hat) {
                if (const int rc = launch_finish_dq(a, medians, y_hat, n_cdfs, (int)index_div, n_blocks, s, ev_dq_begin, ev_dq_end)) return rc;
            }
            if (symbols_out) {
                hipLaunchKernelGGL(rans_dec_finish_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, a);
                SC2_CHECK_LAUNCH();
            }
        }
        return SC2_OK;
    }
    const bool ragged2 = sc2_pol().rans_ragged2 != 0;   // (0: A/B)
    if (indexes && n_entries > 12288 && n_cdfs <= kRagged2Rows && ragged2) {
        // waves (of 16 streams) per workgroup behind one copy of the tables: 1 is the fastest for a launch that runs alone (19.4 /
        // 19.5 / 20.9 / 26.0 ms at 1 / 2 / 4 / 8, round 4); a wide launch inside a pipeline pins 120 KB of LDS per workgroup
        // for ~20 ms, and there two waves per workgroup win (half the CUs held: `bench.py --workload mshp224` 36.6 -> 39.3 k
        // images/s at 2 048 streams per launch, tools/attic/mshp_sweep.sh).  Policy 0 = by stream count.
        const int waves = [n_streams] { const int v = sc2_pol().rans_ragged2_waves; return v <= 0 ? (n_streams >= 1024 ? 2 : 1) : v <= 1 ? 1 : v <= 2 ? 2 : v <= 4 ? 4 : 8; }();
        const size_t lds = (size_t)kRagged2Cap * 2 + (((size_t)kRagged2Rows * 257 * 2 + 15) & ~(size_t)15) + (size_t)kWin * 16 * waves * 4 +
                           (size_t)(3 * kRagged2Rows + 1) * 4 + 16;
        auto kern = waves == 1 ? rans_dec_ragged2_kernel<1> : waves == 2 ? rans_dec_ragged2_kernel<2> : waves == 4 ? rans_dec_ragged2_kernel<4> : rans_dec_ragged2_kernel<8>;
        allow_big_lds(kern, lds);
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            SC2_REQUIRE(gx < (1ll << 31) && n_blocks <= 65535, SC2_ERR_UNSUPPORTED, "rans_decode: problem too large");
            hipLaunchKernelGGL(rans_transpose_in_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, indexes, n_streams,
                               (long long)n_sym, a.ws);
            SC2_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(kern, dim3((n_blocks * 64 + 16 * waves - 1) / (16 * waves)), dim3(64 * waves), lds, s, a);
        SC2_CHECK_LAUNCH();
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            hipLaunchKernelGGL(rans_dec_finish_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, a);
            SC2_CHECK_LAUNCH();
        }
        return SC2_OK;
    }
    if (indexes && n_entries > 12288 && n_cdfs <= kRaggedRows) {
        const int bits = n_cdfs <= 64 ? 8 : 5;   // 257 / 33 bucket bounds per row: 33 KB / 17 KB at the largest row count
        const size_t lds = (size_t)kRaggedCap * 2 + (size_t)(3 * n_cdfs + 1) * 4 +
                           (size_t)n_cdfs * ((1 << bits) + 1) * 2 + 16;
        if (bits == 8) allow_big_lds(rans_dec_ragged_kernel<8>, lds);
        else allow_big_lds(rans_dec_ragged_kernel<5>, lds);
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            SC2_REQUIRE(gx < (1ll << 31) && n_blocks <= 65535, SC2_ERR_UNSUPPORTED, "rans_decode: problem too large");
            hipLaunchKernelGGL(rans_transpose_in_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, indexes, n_streams,
                               (long long)n_sym, a.ws);
            SC2_CHECK_LAUNCH();
        }
        if (bits == 8) hipLaunchKernelGGL(rans_dec_ragged_kernel<8>, dim3(n_blocks), dim3(64), lds, s, a);
        else hipLaunchKernelGGL(rans_dec_ragged_kernel<5>, dim3(n_blocks), dim3(64), lds, s, a);
        SC2_CHECK_LAUNCH();
        if (n_sym > 0) {
            const long long gx = (n_sym + 63) / 64;
            hipLaunchKernelGGL(rans_dec_finish_kernel, dim3((unsigned)gx, n_blocks), dim3(256), 0, s, a);
            SC2_CHECK_LAUNCH();
        }
        return SC2_OK;
    } else if (n_entries <= 12288) {
        hipLaunchKernelGGL(rans_dec_generic_kernel<true>, dim3(n_blocks), dim3(64), (size_t)n_entries * 4, s, a);
    } else {
        hipLaunchKernelGGL(rans_dec_generic_kernel<false>, dim3(n_blocks), dim3(64), 0, s, a);
    }
    SC2_CHECK_LAUNCH();
    return SC2_OK;
}

extern "C" int sc2_rans_decode_batch(const uint8_t *in, int64_t in_stride, const int32_t *in_offset,
                                     const int32_t *in_nbytes, const int32_t *indexes, int64_t index_div,
                                     int n_streams, int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                     const int32_t *cdf_sizes, const int32_t *offsets, int32_t *symbols_out,
                                     int32_t *status, void *workspace, int64_t workspace_bytes, void *stream) {
    return decode_impl(in, in_stride, in_offset, in_nbytes, indexes, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes,
                       offsets, symbols_out, status, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int sc2_rans_decode_dequantize_batch(const uint8_t *in, int64_t in_stride, const int32_t *in_offset,
                                                const int32_t *in_nbytes, int64_t index_div, int n_streams, int64_t n_sym,
                                                const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                                const int32_t *offsets, const float *medians, int32_t *symbols_out,
                                                void *y_hat_bf16_nhwc, int32_t *status, void *workspace, int64_t workspace_bytes,
                                                void *stream) {
    SC2_REQUIRE(y_hat_bf16_nhwc, SC2_ERR_INVALID_ARG, "rans_decode_dequantize: null output");
    return decode_impl(in, in_stride, in_offset, in_nbytes, nullptr, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes,
                       offsets, symbols_out, status, workspace, workspace_bytes, stream, medians, y_hat_bf16_nhwc);
}

// The same call with two caller-owned hipEvent_t recorded on `stream` directly around its last pass (dequantise + NHWC
// transposition): that pass is EntropyModel.dequantize of the reference's decode (sc2bench/models/layer.py:520) and belongs to
// the bottleneck forward, not to the serial coder -- bench.py times it with these events.  Either may be NULL.
extern "C" int sc2_rans_decode_dequantize_batch_ev(const uint8_t *in, int64_t in_stride, const int32_t *in_offset,
                                                   const int32_t *in_nbytes, int64_t index_div, int n_streams, int64_t n_sym,
                                                   const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                                   const int32_t *offsets, const float *medians, int32_t *symbols_out,
                                                   void *y_hat_bf16_nhwc, int32_t *status, void *workspace, int64_t workspace_bytes,
                                                   void *stream, void *ev_dequantize_begin, void *ev_dequantize_end) {
    SC2_REQUIRE(y_hat_bf16_nhwc, SC2_ERR_INVALID_ARG, "rans_decode_dequantize: null output");
    return decode_impl(in, in_stride, in_offset, in_nbytes, nullptr, index_div, n_streams, n_sym, cdfs, n_cdfs, cdf_stride, cdf_sizes,
                       offsets, symbols_out, status, workspace, workspace_bytes, stream, medians, y_hat_bf16_nhwc,
                       ev_dequantize_begin, ev_dequantize_end);
}
