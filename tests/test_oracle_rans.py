"""Oracle (CPU) integer coder vs committed known-answer vectors and vs its independent pure-Python twin."""
import json
import os
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import rans, rans_py

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'rans_kat.json')


@pytest.fixture(scope='module')
def kat():
    with open(GOLDEN) as f:
        return json.load(f)


def test_cdf_known_answers(kat):
    for case in kat['cdf_cases']:
        assert [int(v) for v in rans.pmf_to_quantized_cdf(case['pmf'])] == case['cdf']
        assert rans_py.pmf_to_quantized_cdf(case['pmf']) == case['cdf']


def test_cdf_invariants_and_errors():
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 17, 64, 300):
        p = rng.rand(n).astype(np.float32) ** 4
        p /= p.sum()
        cdf = rans.pmf_to_quantized_cdf(p)
        assert cdf[0] == 0 and cdf[-1] == 65536 and np.all(np.diff(cdf.astype(np.int64)) > 0)
        assert [int(v) for v in cdf] == rans_py.pmf_to_quantized_cdf(p.tolist())
    with pytest.raises(ValueError):
        rans.pmf_to_quantized_cdf([0.5, -0.1, 0.6])
    with pytest.raises(ValueError):
        rans.pmf_to_quantized_cdf([0.5, float('nan')])
    with pytest.raises(ValueError):
        rans.pmf_to_quantized_cdf([0.0, 0.0])
    with pytest.raises(ValueError):
        rans_py.pmf_to_quantized_cdf([0.5, float('inf')])


def test_rans_known_answers(kat):
    t = kat['table']
    for case in kat['cases']:
        enc = rans.encode_with_indexes(case['symbols'], case['indexes'], t['cdfs'], t['cdf_sizes'], t['offsets'])
        assert enc.hex() == case['hex']
        assert rans_py.encode_with_indexes(case['symbols'], case['indexes'], t['cdfs'], t['cdf_sizes'],
                                           t['offsets']).hex() == case['hex']
        dec = rans.decode_with_indexes(bytes.fromhex(case['hex']), case['indexes'], t['cdfs'], t['cdf_sizes'],
                                       t['offsets'])
        assert list(dec) == case['symbols']
        assert len(enc) % 4 == 0 and len(enc) >= 8


def test_rans_two_row_table(kat):
    t, c = kat['table2'], kat['case2']
    rng = random.Random(c['seed'])
    syms = [rng.randint(-6, 6) for _ in range(c['n'])]
    idx = [i % 2 for i in range(c['n'])]
    enc = rans.encode_with_indexes(syms, idx, t['cdfs'], t['cdf_sizes'], t['offsets'])
    assert len(enc) == c['nbytes'] and enc[:32].hex() == c['sha_prefix_hex'] and enc[-16:].hex() == c['tail_hex']
    assert list(rans.decode_with_indexes(enc, idx, t['cdfs'], t['cdf_sizes'], t['offsets'])) == syms
    assert rans_py.decode_with_indexes(enc, idx, t['cdfs'], t['cdf_sizes'], t['offsets']) == syms


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(min_value=-300, max_value=300), min_size=0, max_size=400), st.integers(0, 2 ** 31))
def test_rans_roundtrip_property(symbols, seed):
    rng = np.random.RandomState(seed % (2 ** 31))
    n_rows = 3
    rows, sizes, offs = [], [], []
    for r in range(n_rows):
        n = rng.randint(2, 40)
        p = rng.rand(n).astype(np.float32) + 1e-4
        p /= p.sum()
        cdf = [int(v) for v in rans.pmf_to_quantized_cdf(p)]
        rows.append(cdf)
        sizes.append(len(cdf))
        offs.append(-rng.randint(0, n))
    width = max(len(r) for r in rows)
    rows = [r + [0] * (width - len(r)) for r in rows]
    idx = [int(v) for v in rng.randint(0, n_rows, size=len(symbols))]
    enc = rans.encode_with_indexes(symbols, idx, rows, sizes, offs)
    assert enc == rans_py.encode_with_indexes(symbols, idx, rows, sizes, offs)
    assert list(rans.decode_with_indexes(enc, idx, rows, sizes, offs)) == symbols
