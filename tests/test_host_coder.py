"""The library's HOST range coder (csrc/rans_host.cpp, product code behind sc2_rans_encode_host / sc2_rans_decode_host) against
the committed known-answer vectors and against the oracle's coder: byte-exact streams, exact round trips, status bits.
A host function like the CDF quantiser: no device needed."""
import json
import os
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import rans as oracle_rans

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'rans_kat.json')


def _pad(rows):
    stride = max(len(r) for r in rows)
    cdf = np.zeros((len(rows), stride), np.int32)
    for i, r in enumerate(rows):
        cdf[i, :len(r)] = r
    return cdf


def _random_tables(rng, n_rows, lo=2, hi=40, power=3):
    rows, sizes, offs = [], [], []
    for _ in range(n_rows):
        n = rng.randint(lo, hi)
        p = rng.rand(n).astype(np.float32) ** power + 1e-6
        p /= p.sum()
        c = [int(v) for v in oracle_rans.pmf_to_quantized_cdf(p)]
        rows.append(c)
        sizes.append(len(c))
        offs.append(-(n // 2))
    return _pad(rows), np.array(sizes, np.int32), np.array(offs, np.int32)


def test_known_answers(S):
    kat = json.load(open(GOLDEN))
    t = kat['table']
    tables = S.hip.HostRansTables(_pad(t['cdfs']), t['cdf_sizes'], t['offsets'])
    for case in kat['cases']:
        sym = np.array([case['symbols']], np.int32).reshape(1, -1)
        idx = np.array([case['indexes']], np.int32).reshape(1, -1)
        strings, status = S.hip.rans_encode_host(tables, sym, indexes=idx)
        assert strings[0].hex() == case['hex'] and int(status[0]) == 0
        dec, status = S.hip.rans_decode_host(tables, [bytes.fromhex(case['hex'])], sym.shape[1], indexes=idx)
        assert dec[0].tolist() == case['symbols'] and int(status[0]) == 0
    # SURVEY 8(c) vector 1: the empty stream is the flush of x = 2^31
    strings, _ = S.hip.rans_encode_host(tables, np.zeros((1, 0), np.int32), index_div=1)
    assert strings[0].hex() == '0000008000000000'
    t2, c2 = kat['table2'], kat['case2']
    rng = random.Random(c2['seed'])
    syms = [rng.randint(-6, 6) for _ in range(c2['n'])]
    idx = [i % 2 for i in range(c2['n'])]
    tab2 = S.hip.HostRansTables(_pad(t2['cdfs']), t2['cdf_sizes'], t2['offsets'])
    enc, _ = S.hip.rans_encode_host(tab2, np.array([syms], np.int32), indexes=np.array([idx], np.int32))
    assert len(enc[0]) == c2['nbytes'] and enc[0][:32].hex() == c2['sha_prefix_hex'] and enc[0][-16:].hex() == c2['tail_hex']


def test_matches_oracle_implicit_indexes_with_escapes(S):
    """the entropy bottleneck's layout: row = position // (h * w), a few out-of-range values of both signs and sizes."""
    rng = np.random.RandomState(3)
    cdf, sizes, offs = _random_tables(rng, 24, lo=6, hi=24)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    hw = 11 * 13
    n_sym = 24 * hw
    sym = np.round(rng.randn(5, n_sym) * 3).astype(np.int32)
    sym[0, 5], sym[1, 7], sym[2, 100], sym[3, n_sym - 1], sym[4, 0] = 1000, -100000, (1 << 27) - 3, -77, 15 * 16 ** 3
    strings, status = S.hip.rans_encode_host(tables, sym, index_div=hw)
    idx = (np.arange(n_sym) // hw).astype(np.int32)
    for i in range(sym.shape[0]):
        assert strings[i] == oracle_rans.encode_with_indexes(sym[i], idx, cdf, sizes, offs)
    assert not status.any()
    dec, status = S.hip.rans_decode_host(tables, strings, n_sym, index_div=hw)
    assert np.array_equal(dec, sym) and not status.any()
    # the oracle's decoder reads the host coder's bytes, and the other way round
    assert list(oracle_rans.decode_with_indexes(strings[1], idx, cdf, sizes, offs)) == sym[1].tolist()


def test_matches_oracle_explicit_indexes_wide_table(S):
    """the Gaussian-conditional layout: a per-symbol row index over a table with long rows."""
    rng = np.random.RandomState(5)
    cdf, sizes, offs = _random_tables(rng, 16, lo=100, hi=900, power=6)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    idx = rng.randint(0, 16, size=(3, 4000)).astype(np.int32)
    sym = np.round(rng.randn(3, 4000) * 40).astype(np.int32)
    strings, status = S.hip.rans_encode_host(tables, sym, indexes=idx)
    for i in range(3):
        assert strings[i] == oracle_rans.encode_with_indexes(sym[i], idx[i], cdf, sizes, offs)
    dec, _ = S.hip.rans_decode_host(tables, strings, 4000, indexes=idx)
    assert np.array_equal(dec, sym) and not status.any()


def test_status_bits_and_errors(S):
    rng = np.random.RandomState(7)
    cdf, sizes, offs = _random_tables(rng, 2)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    big = np.zeros((1, 10), np.int32)
    big[0, 3], big[0, 4] = 2 ** 31 - 1, -2 ** 31       # saturated symbols: clamped, status bit 1, no wrap-around
    strings, status = S.hip.rans_encode_host(tables, big, index_div=5)
    assert int(status[0]) == 2
    dec, _ = S.hip.rans_decode_host(tables, strings, 10, index_div=5)
    assert dec[0, 3] == int(sizes[0]) - 2 + (1 << 30) + int(offs[0]) and dec[0, 4] == -(1 << 30) + int(offs[0])
    # a row too small for its stream: status bit 0 (the wrappers then retry with sc2_rans_max_bytes)
    sym = np.round(rng.randn(1, 2000) * 50).astype(np.int32)
    _, status = S.hip.rans_encode_host(tables, sym, index_div=1000, out_stride=64)
    assert int(status[0]) & 1
    strings, status = S.hip.rans_encode_host(tables, sym, index_div=1000, out_stride=S.hip.rans_max_bytes(2000))
    assert int(status[0]) == 0 and np.array_equal(S.hip.rans_decode_host(tables, strings, 2000, index_div=1000)[0], sym)
    # an index outside the table: bit 2
    _, status = S.hip.rans_encode_host(tables, np.zeros((1, 4), np.int32), indexes=np.array([[0, 1, 2, 0]], np.int32))
    assert int(status[0]) & 4
    bad = cdf.copy()
    bad[0, 1] = bad[0, 2]      # a zero-frequency entry
    with pytest.raises(ValueError):
        S.hip.HostRansTables(bad, sizes, offs)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=-300, max_value=300), min_size=0, max_size=300), st.integers(0, 2 ** 31 - 1))
def test_roundtrip_and_oracle_property(S, symbols, seed):
    rng = np.random.RandomState(seed)
    cdf, sizes, offs = _random_tables(rng, 3)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    n = len(symbols)
    idx = rng.randint(0, 3, size=(1, n)).astype(np.int32)
    sym = np.array([symbols], np.int32).reshape(1, n)
    strings, status = S.hip.rans_encode_host(tables, sym, indexes=idx)
    assert strings[0] == oracle_rans.encode_with_indexes(sym[0], idx[0], cdf, sizes, offs) and int(status[0]) == 0
    dec, _ = S.hip.rans_decode_host(tables, strings, n, indexes=idx)
    assert dec[0].tolist() == symbols


def test_many_streams_threaded(S):
    rng = np.random.RandomState(11)
    cdf, sizes, offs = _random_tables(rng, 4, lo=8, hi=20)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    sym = np.round(rng.randn(13, 4 * 50) * 3).astype(np.int32)
    strings, _ = S.hip.rans_encode_host(tables, sym, index_div=50)
    idx = (np.arange(200) // 50).astype(np.int32)
    assert all(strings[i] == oracle_rans.encode_with_indexes(sym[i], idx, cdf, sizes, offs) for i in range(13))
    assert np.array_equal(S.hip.rans_decode_host(tables, strings, 200, index_div=50)[0], sym)


def test_host_decoder_terminates_on_hostile_streams(S):
    """an all-0xFF stream (every escape nibble 0xF: upstream's count loop never ends) and a stream cut short both
    decode to the end, flagged with status bit 3; the API raises on them."""
    rng = np.random.RandomState(5)
    cdf, sizes, offs = _random_tables(rng, 6, lo=6, hi=24)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    n_sym = 600
    sym = np.round(rng.randn(3, n_sym) * 3).astype(np.int32)
    strings, status = S.hip.rans_encode_host(tables, sym, index_div=100)
    assert not status.any()
    hostile = [b'\xff' * len(strings[0]), b'\xff' * 4096, strings[2][:16], strings[1]]
    dec, status = S.hip.rans_decode_host(tables, hostile, n_sym, index_div=100)
    assert all(int(status[i]) & 8 for i in range(3)) and int(status[3]) == 0
    assert np.array_equal(dec[3], sym[1])


def test_reciprocal_division_is_exact(S):
    """The host encoder's quotient (multiplication by the entry's reciprocal, csrc/rans_host.cpp EncEntry) == floor(x / f) for EVERY
    frequency 1 .. 65 536 at the edges of the coder's state range [2^31, 2^63) and around multiples of f, plus random states."""
    L = S.hip.lib()
    rng = random.Random(7)
    edge = [1 << 31, (1 << 31) + 1, (1 << 47) - 1, 1 << 47, (1 << 62) + 12345, (1 << 63) - 1, (1 << 63) - 65536]
    for f in list(range(1, 65537, 1 if os.environ.get('SC2_SLOW_TESTS') else 37)) + [1, 2, 3, 255, 256, 257, 32767, 32768, 32769, 65535, 65536]:
        xs = edge + [rng.randrange(1 << 31, 1 << 63) for _ in range(3)]
        k = rng.randrange(1 << 31, 1 << 63) // f
        xs += [k * f, k * f + f - 1, max(1 << 31, k * f - 1)]
        for x in xs:
            assert L.sc2_rans_host_rcp_div(x, f) == x // f, (x, f)


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 97), st.integers(0, 400))
def test_run_wise_paths_match_oracle(S, seed, index_div, n_sym):
    """The round-6 fast paths of the implicit-index layout -- reciprocal division in the encoder, the one-load 2 048-bucket decode
    index, whole runs of one row -- on PEAKED tables full of frequency-1 entries (the quantiser's floor: the f = 1 special case)
    with a ragged last run: bytes equal to the oracle's, exact round trip; a stream longer than rows x index_div reports bit 2."""
    rng = np.random.RandomState(seed)
    n_rows = 5
    rows, sizes, offs = [], [], []
    for _ in range(n_rows):
        n = rng.randint(3, 60)
        p = np.full(n, 1e-9, np.float64)
        p[rng.randint(0, n)] = 0.9
        p[rng.randint(0, n)] += 0.1
        c = [int(v) for v in oracle_rans.pmf_to_quantized_cdf((p / p.sum()).astype(np.float32))]
        rows.append(c); sizes.append(len(c)); offs.append(-(n // 2))
    cdf, sizes, offs = _pad(rows), np.array(sizes, np.int32), np.array(offs, np.int32)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    n_sym = min(n_sym, n_rows * index_div)
    sym = rng.randint(-35, 36, size=(2, n_sym)).astype(np.int32)
    idx = (np.arange(n_sym) // index_div).astype(np.int32)
    strings, status = S.hip.rans_encode_host(tables, sym, index_div=index_div, out_stride=S.hip.rans_max_bytes(max(1, n_sym)))
    for i in range(2):
        assert strings[i] == oracle_rans.encode_with_indexes(sym[i], idx, cdf, sizes, offs)
    dec, st2 = S.hip.rans_decode_host(tables, strings, n_sym, index_div=index_div)
    assert np.array_equal(dec, sym) and not status.any() and not st2.any()
    too_long = np.zeros((1, n_rows * index_div + 3), np.int32)
    s3, st3 = S.hip.rans_encode_host(tables, too_long, index_div=index_div)
    assert int(st3[0]) & 4
    _, st4 = S.hip.rans_decode_host(tables, s3, too_long.shape[1], index_div=index_div)
    assert int(st4[0]) & 4


def test_code_host_is_encode_then_decode(S):
    """sc2_rans_code_host (both passes per stream in one call: the pipeline's host batches) == sc2_rans_encode_host followed by
    sc2_rans_decode_host: same byte rows, offsets, sizes, decoded symbols and status bits, with and without kept scratch rows,
    including a row too small for its stream (bit 0) and a saturated symbol (bit 1)."""
    rng = np.random.RandomState(21)
    cdf, sizes, offs = _random_tables(rng, 6, lo=5, hi=30)
    tables = S.hip.HostRansTables(cdf, sizes, offs)
    hw = 37
    sym = np.ascontiguousarray(np.round(rng.randn(9, 6 * hw) * 4).astype(np.int32))
    sym[2, 11] = 2 ** 31 - 1
    strings, st_enc = S.hip.rans_encode_host(tables, sym, index_div=hw)
    dec_ref, st_dec = S.hip.rans_decode_host(tables, strings, sym.shape[1], index_div=hw)
    scratch = {}
    for _ in range(2):      # second call: the kept rows
        dec = np.empty_like(sym)
        buf, off, nb, st = S.hip.rans_code_host(tables, sym, hw, dec, scratch=scratch, threads=3)
        assert [buf[i, int(off[i]):int(off[i]) + int(nb[i])].tobytes() for i in range(9)] == strings
        assert np.array_equal(dec, dec_ref) and np.array_equal(st, st_enc | st_dec) and int(st[2]) == 2
    dec = np.empty_like(sym)
    _, _, _, st_small = S.hip.rans_code_host(tables, sym, hw, dec, out_stride=32)
    assert all(int(v) & 1 for v in st_small)
