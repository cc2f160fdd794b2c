// Host-side range coder of libsc2amd.so: the coder the product uses when there are only a few streams to code.
//
// Why it exists: the reference evaluates at test batch size 1 (script/task/image_classification.py:106-145; YAML
// `test_data_loader.batch_size: 1`), i.e. ONE rANS stream per forward.  A range coder is a serial state machine per stream;
// on the GPU one stream is one lane of one wave at ~120-140 ns per symbol (8 + 8.7 ms for the 72 600 symbols of a 224 x 224
// image), while a CPU core steps the same chain in a few ns per symbol.  Below `SC2_HOST_CODER_MAX_STREAMS` streams
// `EntropyBottleneck.compress / decompress` (entropy.py) therefore call these entry points; the batched device coder
// (rans.hip) takes over where its parallelism over streams pays.  Same bit-exact format either way:
// CompressAI's RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes (sc2bench/models/layer.py:506,520):
// rANS with a 64-bit state, lower bound 2^31, 32-bit renormalisation words, 16-bit probabilities, out-of-range values escaped
// through the last CDF entry and coded as 4-bit bypass nibbles.
//
// This is product code with its own structure (nothing here links to or includes anything under oracle/):
//   * tables are prepared ONCE per model into an opaque handle (sc2_rans_host_tables_create): per CDF row the packed
//     cumulative frequencies and a 256-bucket index over them (L1-resident), so the decoder's symbol search is one load plus
//     a scan of 0-2 entries instead of upstream's linear scan from the start of the row;
//   * the encoder walks the symbols BACKWARDS and feeds the state directly (rANS codes in reverse); no intermediate list of
//     pushed symbols is built; words are written back to front into the END of the caller's row, which is exactly the
//     layout of the device coder's output (include/sc2_bottleneck.h: streams are end-aligned in their rows);
//   * streams are independent: they are spread over `n_threads` host threads.
#include "../../include/sc2_bottleneck.h"

#include <cstdint>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

void sc2_set_error(const char *fmt, ...);   // abi.cpp

namespace {

constexpr int kPrecision = 16;
constexpr uint64_t kLower = 1ull << 31;          // rANS64 lower bound of the normalised interval
constexpr int kBypassBits = 4;
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;

// What the ENCODER needs of one table entry (round 6).  The rANS step x' = floor(x / f) 2^16 + x mod f + start is a dependent
// chain through a 64-bit division (25 - 40 cycles on the host cores; 7 ns per symbol measured, 0.51 ms for the 72 600 symbols of
// one 224 x 224 image = a quarter of a bs-1 evaluation forward).  f is a table constant, so the quotient comes from a
// multiplication by its reciprocal (Alverson, "Integer division using reciprocals"): with s = ceil(log2 f) and
// m = ceil(2^(63 + s) / f) < 2^64,  floor(x / f) = mulhi64(x, m) >> (s - 1)  EXACTLY for every x < 2^63 -- the error term
// x (m f - 2^(63 + s)) / (f 2^(63 + s)) is below x 2^s / (f 2^(63 + s)) < 1 / f -- and the coder's state never leaves
// [2^31, 2^63).  f = 1 (s = 0) has no such form: rcp = 0 marks it and the step is x' = x 2^16 + start.
// tests/test_host_coder.py::test_reciprocal_division_is_exact checks the quotient against `/` over every f and edge-case x.
struct EncEntry {
    uint64_t rcp;        // m, or 0 for f = 1
    uint32_t start;
    uint32_t freq_shift; // f in bits 0 .. 16, s - 1 in bits 24 .. 28
};

struct Row {
    int32_t n_entries = 0;   // cdf_size - 1 coded entries; the last one (index max_value) is the escape entry
    int32_t max_value = 0;   // cdf_size - 2
    int32_t offset = 0;
    const uint32_t *cum = nullptr;    // the row's cumulative frequencies: entry k covers [cum[k], cum[k + 1])
    const uint16_t *bucket = nullptr; // [256]: first entry that reaches into cumulative frequencies [256 b, 256 b + 256)
    const EncEntry *enc = nullptr;    // [n_entries], or null when the table is too large to expand (division path)
    const struct FastBucket *fast = nullptr;   // [2048]: one-load decode index for streams coded in whole runs of one row
};

// Decode index at 32-count resolution (round 6): the entry, its start and its frequency in ONE 8-byte load when the bucket
// [32 b, 32 b + 32) lies inside a single entry -- the chain of a step is then load, multiply, add, renormalise; a bucket that an
// entry boundary cuts (a few per cent of the probability mass of a peaked prior) is flagged and searched as before.  16 KB per
// row: L1-resident for the run of symbols that share the row (implicit indexes: 3 025 per row at 224 x 224).
struct FastBucket {
    uint32_t start_entry;   // start in bits 0 .. 15, entry in bits 16 .. 31
    uint32_t freq;          // bit 31: ambiguous (entry = the first candidate)
};

inline EncEntry make_enc_entry(uint32_t start, uint32_t freq) {
    EncEntry e;
    e.start = start;
    if (freq <= 1u) {
        e.rcp = 0;
        e.freq_shift = 1u;
        return e;
    }
    uint32_t s = 0;
    while (freq > (1u << s)) ++s;                                  // s = ceil(log2 f) >= 1
    const unsigned __int128 num = (unsigned __int128)1 << (63 + s);
    const unsigned __int128 m = (num + freq - 1) / freq;           // ceil; < 2^64 because f > 2^(s - 1)
    e.rcp = (uint64_t)m;
    e.freq_shift = freq | ((s - 1) << 24);
    return e;
}

}  // namespace

struct sc2_rans_host_tables {
    std::vector<Row> rows;
    std::vector<uint32_t> cum;
    std::vector<uint16_t> bucket;
    std::vector<EncEntry> enc;
    std::vector<FastBucket> fast;
};

namespace {

// ---- encoder ------------------------------------------------------------------------------------------------------
struct Emitter {
    uint32_t *ptr;          // next free word is ptr[-1]
    uint32_t *limit;        // lowest address a renormalisation word may take (two words below stay free for the flush)
    uint64_t x = kLower;
    int overflow = 0;

    inline void word(uint32_t w) {
        if (ptr > limit) *--ptr = w; else overflow |= 1;
    }
    // one probability-coded entry: x' = floor(x / f) * 2^16 + x mod f + start, after moving 32 bits out if x' would leave
    // [2^31, 2^63)
    inline void put(uint32_t start, uint32_t freq) {
        const uint64_t x_max = ((kLower >> kPrecision) << 32) * (uint64_t)freq;
        if (x >= x_max) { word((uint32_t)x); x >>= 32; }
        x = ((x / freq) << kPrecision) + (x % freq) + start;
    }
    // the same step with the entry's reciprocal (EncEntry): no division, and the renormalisation -- due every 32 / bits-per-symbol
    // symbols, i.e. unpredictably -- without a branch: the word is stored speculatively below the cursor (ptr >= limit = row + 2,
    // so ptr[-1] is inside the row) and the cursor moves only if it was due
    inline void put(const EncEntry &en) {
        const uint32_t freq = en.freq_shift & 0x1FFFFu;
        const uint64_t x_max = (uint64_t)freq << (31 - kPrecision + 32);
        const bool need = x >= x_max, room = ptr > limit;
        ptr[-1] = (uint32_t)x;
        ptr -= (need & room) ? 1 : 0;
        overflow |= (need & !room) ? 1 : 0;
        x = need ? x >> 32 : x;
        if (en.rcp == 0) {      // f = 1 (rare in data: the quantiser's floor frequency)
            x = (x << kPrecision) + en.start;
            return;
        }
        const uint64_t q = (uint64_t)(((unsigned __int128)x * en.rcp) >> 64) >> (en.freq_shift >> 24);
        x = (q << kPrecision) + (x - q * freq) + en.start;
    }
    // one raw nibble: a uniform 4-bit symbol (freq = 2^12 of 2^16)
    inline void put_nibble(uint32_t v) {
        const uint64_t x_max = ((kLower >> kPrecision) << 32) << (kPrecision - kBypassBits);
        if (x >= x_max) { word((uint32_t)x); x >>= 32; }
        x = (x << kBypassBits) | v;
    }
};

void encode_stream(const sc2_rans_host_tables &t, const int32_t *sym, const int32_t *idx, int64_t index_div, int64_t n_sym,
                   uint8_t *row_bytes, int64_t out_stride, int32_t *out_offset, int32_t *out_nbytes, int32_t *status) {
    uint32_t *row = reinterpret_cast<uint32_t *>(row_bytes);
    const int64_t row_words = out_stride / 4;
    Emitter e;
    e.ptr = row + row_words;
    e.limit = row + 2;
    const int n_rows = (int)t.rows.size();
    // one symbol with its row's constants
    auto put_symbol = [&e](const Row &r, int32_t s) {
        long long v = (long long)s - (long long)r.offset;
        int entry;
        if (v >= 0 && v < r.max_value) {
            entry = (int)v;
        } else {
            // escape: value outside [0, max_value).  Decoding order is  entry(max_value), count digits, raw nibbles low to
            // high;  coding runs backwards, so the nibbles go first (high to low), then the count digits, then the entry.
            if (v < -(1ll << 30)) { v = -(1ll << 30); e.overflow |= 2; }
            if (v - r.max_value > (1ll << 30)) { v = (long long)r.max_value + (1ll << 30); e.overflow |= 2; }
            const uint32_t raw = v < 0 ? (uint32_t)(-2 * v - 1) : (uint32_t)(2 * (v - r.max_value));
            int n_nib = 0;
            while (n_nib < 8 && (raw >> (n_nib * kBypassBits)) != 0) ++n_nib;
            for (int j = n_nib - 1; j >= 0; --j) e.put_nibble((raw >> (j * kBypassBits)) & kBypassMax);
            // the count as base-15 "digits": as many 15s as fit, then the remainder (< 15) which ends the count
            const int n15 = n_nib / (int)kBypassMax, rem = n_nib - n15 * (int)kBypassMax;
            e.put_nibble((uint32_t)rem);
            for (int j = 0; j < n15; ++j) e.put_nibble(kBypassMax);
            entry = r.max_value;
        }
        if (r.enc) {
            e.put(r.enc[entry]);
        } else {
            const uint32_t start = r.cum[entry];
            e.put(start, r.cum[entry + 1] - start);
        }
    };
    if (idx) {
        for (int64_t i = n_sym - 1; i >= 0; --i) {
            const int64_t r64 = (int64_t)idx[i];
            if (r64 < 0 || r64 >= n_rows) { e.overflow |= 4; continue; }      // index outside the table: reported, not coded
            put_symbol(t.rows[(size_t)r64], sym[i]);
        }
    } else if (n_sym > 0) {
        // implicit indexes (row = position / index_div): whole runs of one row, last run first (rANS codes in reverse); the row's
        // constants stay in registers for the run
        for (int64_t run = (n_sym - 1) / index_div; run >= 0; --run) {
            const int64_t lo = run * index_div, hi = lo + index_div < n_sym ? lo + index_div : n_sym;
            if (run >= n_rows) { e.overflow |= 4; continue; }
            const Row r = t.rows[(size_t)run];
            for (int64_t i = hi - 1; i >= lo; --i) put_symbol(r, sym[i]);
        }
    }
    // flush: the state's two halves in front of everything (low word first in memory)
    e.ptr -= 2;
    e.ptr[0] = (uint32_t)e.x;
    e.ptr[1] = (uint32_t)(e.x >> 32);
    *out_offset = (int32_t)((e.ptr - row) * 4);
    *out_nbytes = (int32_t)((row + row_words - e.ptr) * 4);
    *status = e.overflow;
}

// ---- decoder ------------------------------------------------------------------------------------------------------
struct Reader {
    const uint32_t *w, *end;
    uint64_t x;
    bool past = false;    // a word was asked for past the end of the stream (status bit 3, as on the device)
    inline uint32_t next() {      // a truncated stream decodes zeros, as on the device
        if (w < end) return *w++;
        past = true;
        return 0u;
    }
    // branch-free: whether a word is due is data-dependent (every ~32 / bits-per-symbol symbols) and a mispredicted branch
    // costs more than the whole step; both selects compile to conditional moves
    inline void renorm() {
        static const uint32_t zero = 0;
        const bool need = x < kLower;
        const uint32_t *src = w < end ? w : &zero;
        const uint64_t refilled = (x << 32) | *src;
        x = need ? refilled : x;
        past |= need & !(w < end);
        w += (need & (w < end)) ? 1 : 0;
    }
    inline uint32_t nibble() {
        const uint32_t v = (uint32_t)x & kBypassMax;
        x >>= kBypassBits;
        renorm();
        return v;
    }
};

void decode_stream(const sc2_rans_host_tables &t, const uint8_t *in, int32_t nbytes, const int32_t *idx, int64_t index_div,
                   int64_t n_sym, int32_t *out, int32_t *status) {
    Reader rd;
    rd.w = reinterpret_cast<const uint32_t *>(in);
    rd.end = rd.w + nbytes / 4;
    const uint64_t lo = rd.next(), hi = rd.next();
    rd.x = lo | (hi << 32);
    const int n_rows = (int)t.rows.size();
    int st = 0;
    auto get_symbol = [&rd, &st](const Row &r, const bool fine) -> int32_t {
        const uint32_t cf = (uint32_t)rd.x & 0xFFFFu;
        // symbol search: a 256-bucket table per row (the 24 x 512 B of a factorised prior stay in L1; an exact 65 536-entry
        // table per row -- 3 MB for 24 rows -- missed L2 on nearly every symbol, 60 cycles per step) + a short forward scan.
        // Round 6: a stream coded in whole runs of one row (implicit indexes: 3 025 symbols per row at 224 x 224) looks at ONE
        // row for a whole run, so that row's 4 096-bucket index (8 KB) is L1-resident for the run and the scan behind it almost
        // never steps (an entry narrower than 16 counts): a predictable branch instead of a coin flip per symbol
        int entry;
        uint32_t start, freq;
        if (fine) {
            const FastBucket b = r.fast[cf >> 5];
            entry = (int)(b.start_entry >> 16);
            start = b.start_entry & 0xFFFFu;
            freq = b.freq;
            if (__builtin_expect((b.freq >> 31) != 0, 0)) {
                while (r.cum[entry + 1] <= cf) ++entry;
                start = r.cum[entry];
                freq = r.cum[entry + 1] - start;
            }
        } else {
            entry = r.bucket[cf >> 8];
            while (r.cum[entry + 1] <= cf) ++entry;
            start = r.cum[entry];
            freq = r.cum[entry + 1] - start;
        }
        rd.x = (uint64_t)freq * (rd.x >> kPrecision) + cf - start;
        rd.renorm();
        int32_t v = entry;
        if (entry == r.max_value) {
            // the nibble count of a value an encoder escaped is ONE digit <= 8 (a 32-bit raw value); anything else is a corrupt
            // or hostile stream: flagged (bit 3), coded as the escape entry itself -- the device decoders do the same
            const int n_nib = (int)rd.nibble();
            if (n_nib > 8) {
                st |= 8;
            } else {
                uint32_t raw = 0;
                for (int j = 0; j < n_nib; ++j) raw |= rd.nibble() << (j * kBypassBits);
                v = (raw & 1u) ? -(int32_t)(raw >> 1) - 1 : (int32_t)(raw >> 1) + r.max_value;
            }
        }
        return v + r.offset;
    };
    if (idx) {
        for (int64_t i = 0; i < n_sym; ++i) {
            const int64_t r64 = (int64_t)idx[i];
            if (r64 < 0 || r64 >= n_rows) { st |= 4; out[i] = 0; continue; }
            out[i] = get_symbol(t.rows[(size_t)r64], false);
        }
    } else {
        for (int64_t lo = 0, run = 0; lo < n_sym; lo += index_div, ++run) {      // whole runs of one row
            const int64_t hi = lo + index_div < n_sym ? lo + index_div : n_sym;
            if (run >= n_rows) {
                st |= 4;
                for (int64_t i = lo; i < hi; ++i) out[i] = 0;
                continue;
            }
            const Row r = t.rows[(size_t)run];
            if (r.fast) {
                for (int64_t i = lo; i < hi; ++i) out[i] = get_symbol(r, true);
            } else {
                for (int64_t i = lo; i < hi; ++i) out[i] = get_symbol(r, false);
            }
        }
    }
    *status = st | (rd.past ? 8 : 0);
}

template <class F>
void for_streams(int n_streams, int n_threads, F &&f) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_streams) n_threads = n_streams;
    if (n_threads <= 1) {
        for (int s = 0; s < n_streams; ++s) f(s);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_threads);
    for (int k = 0; k < n_threads; ++k)
        pool.emplace_back([&, k]() { for (int s = k; s < n_streams; s += n_threads) f(s); });
    for (auto &th : pool) th.join();
}

}  // namespace

#define HOST_REQUIRE(cond, msg)                 \
    do {                                        \
        if (!(cond)) {                          \
            sc2_set_error("%s", msg);            \
            return SC2_ERR_INVALID_ARG;             \
        }                                       \
    } while (0)

extern "C" int sc2_rans_host_tables_create(const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                           const int32_t *offsets, sc2_rans_host_tables **out) {
    HOST_REQUIRE(cdfs && cdf_sizes && offsets && out, "sc2_rans_host_tables_create: null argument");
    HOST_REQUIRE(n_cdfs > 0 && cdf_stride >= 3, "sc2_rans_host_tables_create: empty table");
    sc2_rans_host_tables *t = new (std::nothrow) sc2_rans_host_tables();
    HOST_REQUIRE(t != nullptr, "sc2_rans_host_tables_create: out of memory");
    t->rows.resize((size_t)n_cdfs);
    t->cum.assign((size_t)n_cdfs * (size_t)cdf_stride, 0u);
    t->bucket.assign((size_t)n_cdfs * 256u, 0);
    // encoder entries (16 B each) for tables up to 4 M entries (64 MB); beyond that the division path
    const bool expand = (size_t)n_cdfs * (size_t)cdf_stride <= ((size_t)1 << 22);
    if (expand) t->enc.assign((size_t)n_cdfs * (size_t)cdf_stride, EncEntry{0, 0, 1u});
    const bool fine = n_cdfs <= 4096 && cdf_stride <= 65536;      // 16 KB per row; entry indexes fit 16 bits
    if (fine) t->fast.assign((size_t)n_cdfs * 2048u, FastBucket{0u, 0u});
    for (int r = 0; r < n_cdfs; ++r) {
        const int32_t *c = cdfs + (size_t)r * cdf_stride;
        const int size = cdf_sizes[r];
        bool ok = size >= 3 && size <= cdf_stride && c[0] == 0 && c[size - 1] == (1 << kPrecision);
        for (int k = 0; ok && k + 1 < size; ++k) ok = c[k + 1] > c[k];      // every coded entry has a non-zero frequency
        if (!ok) {
            delete t;
            sc2_set_error("%s", "sc2_rans_host_tables_create: a CDF row is not a strictly increasing 16-bit table of >= 2 entries");
            return SC2_ERR_INVALID_ARG;
        }
        Row &row = t->rows[(size_t)r];
        row.n_entries = size - 1;
        row.max_value = size - 2;
        row.offset = offsets[r];
        uint32_t *cum = t->cum.data() + (size_t)r * cdf_stride;
        uint16_t *bucket = t->bucket.data() + (size_t)r * 256u;
        for (int k = 0; k < size; ++k) cum[k] = (uint32_t)c[k];
        for (int b = 0, k = 0; b < 256; ++b) {
            while (cum[k + 1] <= (uint32_t)b * 256u) ++k;     // entry k is the first with cum[k + 1] > 256 b
            bucket[b] = (uint16_t)k;
        }
        row.cum = cum;
        row.bucket = bucket;
        if (fine) {
            FastBucket *fb = t->fast.data() + (size_t)r * 2048u;
            for (int b = 0, k = 0; b < 2048; ++b) {
                while (cum[k + 1] <= (uint32_t)b * 32u) ++k;        // entry k is the first with cum[k + 1] > 32 b
                const bool whole = cum[k + 1] >= (uint32_t)b * 32u + 32u;   // ... and it covers the whole bucket
                fb[b].start_entry = (cum[k] & 0xFFFFu) | ((uint32_t)k << 16);
                fb[b].freq = whole ? cum[k + 1] - cum[k] : 0x80000000u;
            }
            row.fast = fb;
        }
        if (expand) {
            EncEntry *enc = t->enc.data() + (size_t)r * cdf_stride;
            for (int k = 0; k + 1 < size; ++k) enc[k] = make_enc_entry(cum[k], cum[k + 1] - cum[k]);
            row.enc = enc;
        }
    }
    *out = t;
    return SC2_OK;
}

// floor(x / freq) by the encoder's reciprocal form (test hook: tests/test_host_coder.py sweeps every freq against `/`)
extern "C" uint64_t sc2_rans_host_rcp_div(uint64_t x, uint32_t freq) {
    const EncEntry en = make_enc_entry(0u, freq);
    if (en.rcp == 0) return x;
    return (uint64_t)(((unsigned __int128)x * en.rcp) >> 64) >> (en.freq_shift >> 24);
}

extern "C" void sc2_rans_host_tables_destroy(sc2_rans_host_tables *t) { delete t; }

extern "C" int sc2_rans_encode_host(const sc2_rans_host_tables *t, const int32_t *symbols, const int32_t *indexes,
                                    int64_t index_div, int n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride,
                                    int32_t *out_offset, int32_t *out_nbytes, int32_t *status, int n_threads) {
    HOST_REQUIRE(t && out && out_offset && out_nbytes && status && (symbols || n_sym == 0), "sc2_rans_encode_host: null argument");
    HOST_REQUIRE(n_streams >= 0 && n_sym >= 0, "sc2_rans_encode_host: negative size");
    HOST_REQUIRE(out_stride >= 8 && out_stride % 4 == 0, "sc2_rans_encode_host: out_stride must be a multiple of 4, >= 8");
    HOST_REQUIRE(indexes || index_div > 0 || n_sym == 0, "sc2_rans_encode_host: need indexes or index_div");
    HOST_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3u) == 0, "sc2_rans_encode_host: out must be 4-byte aligned");
    for_streams(n_streams, n_threads, [&](int s) {
        encode_stream(*t, symbols + (size_t)s * n_sym, indexes ? indexes + (size_t)s * n_sym : nullptr, index_div, n_sym,
                      out + (size_t)s * out_stride, out_stride, out_offset + s, out_nbytes + s, status + s);
    });
    return SC2_OK;
}

extern "C" int sc2_rans_decode_host(const sc2_rans_host_tables *t, const uint8_t *in, int64_t in_stride,
                                    const int32_t *in_offset, const int32_t *in_nbytes, const int32_t *indexes,
                                    int64_t index_div, int n_streams, int64_t n_sym, int32_t *symbols_out, int32_t *status,
                                    int n_threads) {
    HOST_REQUIRE(t && in && in_offset && in_nbytes && status && (symbols_out || n_sym == 0), "sc2_rans_decode_host: null argument");
    HOST_REQUIRE(n_streams >= 0 && n_sym >= 0, "sc2_rans_decode_host: negative size");
    HOST_REQUIRE(indexes || index_div > 0 || n_sym == 0, "sc2_rans_decode_host: need indexes or index_div");
    HOST_REQUIRE((reinterpret_cast<uintptr_t>(in) & 3u) == 0 && in_stride % 4 == 0, "sc2_rans_decode_host: in / in_stride must be 4-byte aligned");
    for (int s = 0; s < n_streams; ++s)
        HOST_REQUIRE(in_offset[s] >= 0 && in_offset[s] % 4 == 0 && in_nbytes[s] >= 0 &&
                     (int64_t)in_offset[s] + in_nbytes[s] <= in_stride, "sc2_rans_decode_host: stream outside its row");
    for_streams(n_streams, n_threads, [&](int s) {
        decode_stream(*t, in + (size_t)s * in_stride + in_offset[s], in_nbytes[s],
                      indexes ? indexes + (size_t)s * n_sym : nullptr, index_div, n_sym, symbols_out + (size_t)s * n_sym, status + s);
    });
    return SC2_OK;
}

// Encode AND decode every stream on the host in one call (round 6: the first coder groups of a pipelined run, pipeline.py): a
// thread codes a stream into its row and decodes that row straight away -- the row is still in the core's cache, and the two passes
// share one thread start.  Arguments as the two calls above; status = encode status | decode status.
extern "C" int sc2_rans_code_host(const sc2_rans_host_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t index_div,
                                  int n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride, int32_t *out_offset,
                                  int32_t *out_nbytes, int32_t *symbols_out, int32_t *status, int n_threads) {
    HOST_REQUIRE(t && out && out_offset && out_nbytes && status && ((symbols && symbols_out) || n_sym == 0), "sc2_rans_code_host: null argument");
    HOST_REQUIRE(n_streams >= 0 && n_sym >= 0, "sc2_rans_code_host: negative size");
    HOST_REQUIRE(out_stride >= 8 && out_stride % 4 == 0, "sc2_rans_code_host: out_stride must be a multiple of 4, >= 8");
    HOST_REQUIRE(indexes || index_div > 0 || n_sym == 0, "sc2_rans_code_host: need indexes or index_div");
    HOST_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3u) == 0, "sc2_rans_code_host: out must be 4-byte aligned");
    for_streams(n_streams, n_threads, [&](int s) {
        const int32_t *idx = indexes ? indexes + (size_t)s * n_sym : nullptr;
        int32_t st_enc = 0, st_dec = 0;
        encode_stream(*t, symbols + (size_t)s * n_sym, idx, index_div, n_sym, out + (size_t)s * out_stride, out_stride, out_offset + s,
                      out_nbytes + s, &st_enc);
        decode_stream(*t, out + (size_t)s * out_stride + out_offset[s], out_nbytes[s], idx, index_div, n_sym,
                      symbols_out + (size_t)s * n_sym, &st_dec);
        status[s] = st_enc | st_dec;
    });
    return SC2_OK;
}

