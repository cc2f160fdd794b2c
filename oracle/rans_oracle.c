/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the integer entropy-coding arithmetic that the
 * reference (sc2bench) drives through CompressAI:
 *
 *   - pmf_to_quantized_cdf        (compressai/cpp_exts/ops/ops.cpp)
 *   - RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes
 *                                 (compressai/cpp_exts/rans/rans_interface.cpp,
 *                                  third_party/ryg_rans/rans64.h)
 *
 * Reference call sites that reach this arithmetic (file:line under
 * /root/reference): sc2bench/models/layer.py:506 (encode -> compress),
 * layer.py:520 (decode -> decompress), layer.py:371,386 (EntropyBottleneckLayer),
 * layer.py:431-441 (update -> pmf_to_quantized_cdf).
 *
 * PARITY UNPINNED: CompressAI (compressai>=1.2.3, setup.py:28, no lock file)
 * is a third-party dependency that is neither vendored under /root/reference
 * nor installable here, and the reference ships no tests / golden vectors for
 * this path.  This file restates the published
 * algorithm; it is pinned only by self-derived known-answer vectors
 * (tests/golden/rans_kat.json) and by a second, independent pure-Python
 * restatement (oracle/rans_py.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_PRECISION 16
#define ORACLE_BYPASS_PRECISION 4
#define ORACLE_MAX_BYPASS_VAL ((1 << ORACLE_BYPASS_PRECISION) - 1)
#define ORACLE_RANS64_L (1ull << 31)

/* ------------------------------------------------------------------------ */
/* pmf_to_quantized_cdf: float32 pmf[n] -> uint32 cdf[n+1], sum = 2^precision */
/* returns 0 ok, -1 negative/non-finite probability, -2 all-zero pmf,       */
/* -3 no symbol to steal from (upstream assert)                             */
/* ------------------------------------------------------------------------ */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf)
{
    int i, j;
    for (i = 0; i < n; ++i) {
        if (pmf[i] < 0.0f || !isfinite(pmf[i]))
            return -1;
    }
    cdf[0] = 0;
    for (i = 0; i < n; ++i) {
        /* std::round(float * int) : float32 product, half away from zero */
        float scaled = pmf[i] * (float)(1 << precision);
        cdf[i + 1] = (uint32_t)roundf(scaled);
    }
    /* std::accumulate(..., 0) accumulates in int */
    {
        int total_i = 0;
        uint32_t total;
        for (i = 0; i < n + 1; ++i)
            total_i += (int)cdf[i];
        total = (uint32_t)total_i;
        if (total == 0)
            return -2;
        for (i = 0; i < n + 1; ++i)
            cdf[i] = (uint32_t)((((uint64_t)1 << precision) * (uint64_t)cdf[i]) / total);
    }
    for (i = 1; i < n + 1; ++i)
        cdf[i] += cdf[i - 1];
    cdf[n] = (uint32_t)1 << precision;

    for (i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (j = 0; j < n; ++j) {
                uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) {
                    best_freq = freq;
                    best_steal = j;
                }
            }
            if (best_steal == -1)
                return -3;
            if (best_steal < i) {
                for (j = best_steal + 1; j <= i; ++j)
                    cdf[j]--;
            } else {
                for (j = i + 1; j <= best_steal; ++j)
                    cdf[j]++;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* rANS64 encoder                                                           */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint16_t start;
    uint16_t range;
    uint8_t bypass;
} oracle_sym_t;

typedef struct {
    oracle_sym_t *v;
    size_t n, cap;
} oracle_symvec_t;

static int symvec_push(oracle_symvec_t *s, uint16_t start, uint16_t range, int bypass)
{
    if (s->n == s->cap) {
        size_t ncap = s->cap ? s->cap * 2 : 1024;
        oracle_sym_t *nv = (oracle_sym_t *)realloc(s->v, ncap * sizeof(oracle_sym_t));
        if (!nv)
            return -1;
        s->v = nv;
        s->cap = ncap;
    }
    s->v[s->n].start = start;
    s->v[s->n].range = range;
    s->v[s->n].bypass = (uint8_t)bypass;
    s->n++;
    return 0;
}

/*
 * Encodes n symbols (in order) into one rANS stream.
 *   cdfs      : int32 [n_cdfs][cdf_stride] row-major quantized CDF table
 *   cdf_sizes : int32 [n_cdfs]  (= pmf_length + 2)
 *   offsets   : int32 [n_cdfs]
 * Output bytes are written to out[0..ret) ; returns number of bytes,
 * or -1 on allocation failure, -2 if out_cap is too small.
 */
long oracle_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, long n,
                                     const int32_t *cdfs, int cdf_stride,
                                     const int32_t *cdf_sizes, const int32_t *offsets,
                                     uint8_t *out, long out_cap)
{
    oracle_symvec_t syms = {0, 0, 0};
    long i;
    for (i = 0; i < n; ++i) {
        const int32_t cdf_idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)cdf_idx * cdf_stride;
        const int32_t max_value = cdf_sizes[cdf_idx] - 2;
        int32_t value = symbols[i] - offsets[cdf_idx];
        uint32_t raw_val = 0;
        if (value < 0) {
            raw_val = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw_val = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        if (symvec_push(&syms, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0))
            goto oom;
        if (value == max_value) {
            int32_t n_bypass = 0, val, j;
            while ((raw_val >> (n_bypass * ORACLE_BYPASS_PRECISION)) != 0)
                ++n_bypass;
            val = n_bypass;
            while (val >= ORACLE_MAX_BYPASS_VAL) {
                if (symvec_push(&syms, ORACLE_MAX_BYPASS_VAL, ORACLE_MAX_BYPASS_VAL + 1, 1))
                    goto oom;
                val -= ORACLE_MAX_BYPASS_VAL;
            }
            if (symvec_push(&syms, (uint16_t)val, (uint16_t)(val + 1), 1))
                goto oom;
            for (j = 0; j < n_bypass; ++j) {
                const int32_t v4 = (int32_t)((raw_val >> (j * ORACLE_BYPASS_PRECISION)) & ORACLE_MAX_BYPASS_VAL);
                if (symvec_push(&syms, (uint16_t)v4, (uint16_t)(v4 + 1), 1))
                    goto oom;
            }
        }
    }
    {
        /* flush(): walk the pushed symbols backwards, write words backwards */
        const size_t n_words = syms.n + 2; /* upstream sizes by #pushed; +2 keeps the flush in bounds */
        uint32_t *buf = (uint32_t *)malloc(n_words * sizeof(uint32_t));
        uint32_t *ptr;
        uint64_t x = ORACLE_RANS64_L;
        size_t k;
        long nbytes;
        if (!buf)
            goto oom;
        ptr = buf + n_words;
        for (k = syms.n; k-- > 0;) {
            const oracle_sym_t s = syms.v[k];
            if (!s.bypass) {
                const uint32_t freq = s.range;
                const uint64_t x_max = ((ORACLE_RANS64_L >> ORACLE_PRECISION) << 32) * (uint64_t)freq;
                if (x >= x_max) {
                    ptr -= 1;
                    *ptr = (uint32_t)x;
                    x >>= 32;
                }
                x = ((x / freq) << ORACLE_PRECISION) + (x % freq) + s.start;
            } else {
                const uint32_t freq = 1u << (16 - ORACLE_BYPASS_PRECISION);
                const uint64_t x_max = ((ORACLE_RANS64_L >> 16) << 32) * (uint64_t)freq;
                if (x >= x_max) {
                    ptr -= 1;
                    *ptr = (uint32_t)x;
                    x >>= 32;
                }
                x = (x << ORACLE_BYPASS_PRECISION) | s.start;
            }
        }
        ptr -= 2;
        ptr[0] = (uint32_t)(x >> 0);
        ptr[1] = (uint32_t)(x >> 32);
        nbytes = (long)((buf + n_words) - ptr) * (long)sizeof(uint32_t);
        if (nbytes > out_cap) {
            free(buf);
            free(syms.v);
            return -2;
        }
        memcpy(out, ptr, (size_t)nbytes); /* host is little-endian, as upstream assumes */
        free(buf);
        free(syms.v);
        return nbytes;
    }
oom:
    free(syms.v);
    return -1;
}

/* ------------------------------------------------------------------------ */
/* rANS64 decoder                                                           */
/* ------------------------------------------------------------------------ */
static inline uint32_t dec_get_bits(uint64_t *r, const uint32_t **pptr, uint32_t n_bits)
{
    uint64_t x = *r;
    uint32_t val = (uint32_t)(x & ((1u << n_bits) - 1));
    x = x >> n_bits;
    if (x < ORACLE_RANS64_L) {
        x = (x << 32) | **pptr;
        *pptr += 1;
    }
    *r = x;
    return val;
}

int oracle_rans_decode_with_indexes(const uint8_t *encoded, long nbytes,
                                    const int32_t *indexes, long n,
                                    const int32_t *cdfs, int cdf_stride,
                                    const int32_t *cdf_sizes, const int32_t *offsets,
                                    int32_t *out)
{
    /* copy to an aligned word buffer (upstream casts the string's bytes) */
    long n_words = nbytes / 4;
    uint32_t *words = (uint32_t *)malloc((size_t)(n_words + 4) * sizeof(uint32_t));
    const uint32_t *ptr;
    uint64_t x;
    long i;
    if (!words)
        return -1;
    memset(words, 0, (size_t)(n_words + 4) * sizeof(uint32_t));
    memcpy(words, encoded, (size_t)nbytes);
    ptr = words;
    x = (uint64_t)ptr[0];
    x |= (uint64_t)ptr[1] << 32;
    ptr += 2;
    for (i = 0; i < n; ++i) {
        const int32_t cdf_idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)cdf_idx * cdf_stride;
        const int32_t max_value = cdf_sizes[cdf_idx] - 2;
        const int32_t offset = offsets[cdf_idx];
        const uint32_t cum_freq = (uint32_t)(x & ((1u << ORACLE_PRECISION) - 1));
        int32_t k = 0, value;
        uint32_t s, start, freq;
        /* std::find_if(first, first + cdf_size, v > cum_freq) */
        while (k < cdf_sizes[cdf_idx] && !((uint32_t)cdf[k] > cum_freq))
            ++k;
        s = (uint32_t)(k - 1);
        start = (uint32_t)cdf[s];
        freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        /* Rans64DecAdvance */
        x = (uint64_t)freq * (x >> ORACLE_PRECISION) + (x & ((1u << ORACLE_PRECISION) - 1)) - start;
        if (x < ORACLE_RANS64_L) {
            x = (x << 32) | *ptr;
            ptr += 1;
        }
        value = (int32_t)s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(&x, &ptr, ORACLE_BYPASS_PRECISION);
            int32_t n_bypass = val, raw_val = 0, j;
            while (val == ORACLE_MAX_BYPASS_VAL) {
                val = (int32_t)dec_get_bits(&x, &ptr, ORACLE_BYPASS_PRECISION);
                n_bypass += val;
            }
            for (j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(&x, &ptr, ORACLE_BYPASS_PRECISION);
                raw_val |= val << (j * ORACLE_BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1)
                value = -value - 1;
            else
                value += max_value;
        }
        out[i] = value + offset;
    }
    free(words);
    return 0;
}
