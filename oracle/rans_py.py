"""ORACLE -- TEST INFRASTRUCTURE ONLY (pure-Python loops, small cases only).

Second, independent restatement of CompressAI's integer entropy-coding
arithmetic, written directly from the published algorithm (SURVEY.md section
8(a) algorithm box; upstream ``compressai/cpp_exts/ops/ops.cpp``,
``cpp_exts/rans/rans_interface.cpp``, ``third_party/ryg_rans/rans64.h``).
It exists to cross-check ``oracle/rans_oracle.c``: two restatements written in
different languages agreeing byte-for-byte, plus the self-derived known-answer
vectors in ``tests/golden/rans_kat.json``, is the strongest pin available.

PARITY UNPINNED: the reference (/root/reference) holds no tests or golden
vectors for this path, and CompressAI cannot be imported here.

Reference call sites: sc2bench/models/layer.py:506,520 (compress/decompress),
layer.py:431-441 (update).
"""
import math
import struct

PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS_VAL = (1 << BYPASS_PRECISION) - 1
RANS64_L = 1 << 31
_M32 = 0xFFFFFFFF


def _f32(x):
    return struct.unpack('<f', struct.pack('<f', x))[0]


def pmf_to_quantized_cdf(pmf, precision=16):
    """float pmf -> list of uint32 CDF entries (len+1); raises ValueError like upstream's domain_error."""
    pmf32 = [_f32(p) for p in pmf]
    for p in pmf32:
        if p < 0 or not math.isfinite(p):
            raise ValueError('Invalid `pmf`, non-finite or negative element found: {}'.format(p))
    cdf = [0] * (len(pmf32) + 1)
    for i, p in enumerate(pmf32):
        scaled = _f32(p * float(1 << precision))  # float32 product
        # roundf: half away from zero; scaled >= 0 here
        fl = math.floor(scaled)
        cdf[i + 1] = int(fl + 1) if (scaled - fl) >= 0.5 else int(fl)
    total = sum(cdf)
    if total == 0:
        raise ValueError('Invalid `pmf`: at least one element must have a non-zero probability.')
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq = None
            best_steal = -1
            for j in range(n):
                freq = cdf[j + 1] - cdf[j]
                if freq > 1 and (best_freq is None or freq < best_freq):
                    best_freq = freq
                    best_steal = j
            assert best_steal != -1
            if best_steal < i:
                for j in range(best_steal + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best_steal + 1):
                    cdf[j] += 1
    return cdf


def _push_symbols(symbols, indexes, cdfs, cdf_sizes, offsets):
    pushed = []
    for s, idx in zip(symbols, indexes):
        cdf = cdfs[idx]
        max_value = cdf_sizes[idx] - 2
        value = s - offsets[idx]
        raw_val = 0
        if value < 0:
            raw_val = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw_val = 2 * (value - max_value)
            value = max_value
        pushed.append((cdf[value] & 0xFFFF, (cdf[value + 1] - cdf[value]) & 0xFFFF, False))
        if value == max_value:
            n_bypass = 0
            while (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS_VAL:
                pushed.append((MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, True))
                val -= MAX_BYPASS_VAL
            pushed.append((val, val + 1, True))
            for j in range(n_bypass):
                v4 = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL
                pushed.append((v4, v4 + 1, True))
    return pushed


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    """Returns the rANS byte string for one stream (as CompressAI's RansEncoder would)."""
    pushed = _push_symbols(symbols, indexes, cdfs, cdf_sizes, offsets)
    x = RANS64_L
    words = []  # emitted in order; the final layout is reversed
    for start, rng, bypass in reversed(pushed):
        if not bypass:
            x_max = ((RANS64_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & _M32)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS64_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & _M32)
                x >>= 32
            x = (x << BYPASS_PRECISION) | start
    out = [x & _M32, (x >> 32) & _M32] + list(reversed(words))
    return struct.pack('<{}I'.format(len(out)), *out)


def decode_with_indexes(encoded, indexes, cdfs, cdf_sizes, offsets):
    n_words = len(encoded) // 4
    words = list(struct.unpack('<{}I'.format(n_words), encoded[:4 * n_words])) + [0, 0, 0, 0]
    pos = 2
    x = words[0] | (words[1] << 32)

    def get_bits(n_bits):
        nonlocal x, pos
        val = x & ((1 << n_bits) - 1)
        x >>= n_bits
        if x < RANS64_L:
            x = (x << 32) | words[pos]
            pos += 1
        return val

    out = []
    for idx in indexes:
        cdf = cdfs[idx]
        max_value = cdf_sizes[idx] - 2
        cum_freq = x & 0xFFFF
        k = 0
        while k < cdf_sizes[idx] and not (cdf[k] > cum_freq):
            k += 1
        s = k - 1
        start = cdf[s]
        freq = cdf[s + 1] - cdf[s]
        x = freq * (x >> PRECISION) + (x & 0xFFFF) - start
        if x < RANS64_L:
            x = (x << 32) | words[pos]
            pos += 1
        value = s
        if value == max_value:
            val = get_bits(BYPASS_PRECISION)
            n_bypass = val
            while val == MAX_BYPASS_VAL:
                val = get_bits(BYPASS_PRECISION)
                n_bypass += val
            raw_val = 0
            for j in range(n_bypass):
                val = get_bits(BYPASS_PRECISION)
                raw_val |= val << (j * BYPASS_PRECISION)
            value = raw_val >> 1
            if raw_val & 1:
                value = -value - 1
            else:
                value += max_value
        out.append(value + offsets[idx])
    return out
