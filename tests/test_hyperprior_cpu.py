"""CPU tests of the hyperprior rows: the oracle against its committed golden vectors, the
Gaussian-conditional table build of the product module (host code) against the oracle, registry / state-dict /
error behaviour of the SHP / MSHP classes.  No compute call goes to the device here."""
import hashlib
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from recipe import build_oracle_hyperprior, fingerprint  # noqa: E402

NAMES = ('SHPBasedResNetBottleneck', 'MSHPBasedResNetBottleneck')


@pytest.fixture(scope='module')
def golden():
    return torch.load(os.path.join(HERE, 'golden', 'hyperprior_golden.pt'), weights_only=False)


@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_hyperprior_golden(R, golden, name):
    g = golden[name]
    m, x = build_oracle_hyperprior(R, name)
    assert fingerprint(m) == pytest.approx(g['fingerprint'], rel=1e-6)
    assert torch.equal(x, g['x'])
    with torch.no_grad():
        y = m.g_a(x)
        z = m.h_a(torch.abs(y) if name.startswith('SHP') else y)
        torch.testing.assert_close(y, g['y'], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(z, g['z'], rtol=1e-5, atol=1e-5)
        params = m.h_s(g['z_hat'])
        scales, means = (params, None) if name.startswith('SHP') else params.chunk(2, 1)
        y_hat, y_lik = m.gaussian_conditional(g['y'], scales, means=means)
        assert torch.equal(y_hat, g['y_hat'])
        torch.testing.assert_close(y_lik, g['y_lik'], rtol=1e-4, atol=1e-9)
        yn, ln = m.gaussian_conditional(g['y'], scales, means=means, training=True, noise=g['noise_y'])
        assert torch.equal(yn, g['y'] + g['noise_y'])          # noise mode ignores the means (upstream semantics)
        torch.testing.assert_close(ln, g['y_lik_noise'], rtol=1e-4, atol=1e-9)
        m.update()
        gc = m.gaussian_conditional
        assert hashlib.sha256(gc._quantized_cdf.numpy().tobytes()).hexdigest() == g['gc_cdf_sha256']
        assert torch.equal(gc._offset, g['gc_offset']) and torch.equal(gc._cdf_length, g['gc_cdf_length'])
        assert torch.equal(gc.build_indexes(scales), g['indexes'])
        enc = m.encode(x)
        assert [s.hex() for s in enc['strings'][0]] == g['y_strings_hex']
        assert [s.hex() for s in enc['strings'][1]] == g['z_strings_hex']
        assert list(enc['shape']) == g['shape']
        torch.testing.assert_close(m.decode(**enc), g['decoded'], rtol=1e-5, atol=1e-5)
        assert R.file_size(enc) == g['file_size_kb']


def test_gaussian_tables_invariants(R):
    gc = R.GaussianConditional(None)
    assert gc.update_scale_table(R.get_scale_table()) is True
    assert gc.update_scale_table(R.get_scale_table()) is False          # already built, not forced
    table = gc.scale_table
    assert table.numel() == 64 and abs(table[0].item() - 0.11) < 1e-6 and abs(table[-1].item() - 256) < 1e-3
    assert tuple(gc._quantized_cdf.shape) == (64, 3133)                 # 2 * ceil(256 * 6.1094...) + 1 + 2
    for i in range(64):
        n = int(gc._cdf_length[i])
        row = gc._quantized_cdf[i, :n].long()
        center = -int(gc._offset[i])
        assert n == 2 * center + 3 and row[0] == 0 and row[-1] == 65536
        assert bool((row[1:] > row[:-1]).all())                        # every symbol keeps a non-zero frequency
        assert bool((gc._quantized_cdf[i, n:] == 0).all())
    # indexes: monotone in the scale, clamped at the bound, last row for anything above the table
    s = torch.tensor([[0.0, 0.05, 0.11, 0.12, 1.0, 255.9, 256.0, 1e6]])
    idx = gc.build_indexes(s)[0].tolist()
    assert idx[0] == idx[1] == idx[2] == 0 and idx == sorted(idx) and idx[-1] == 63 and idx[-2] == 63


@pytest.mark.parametrize('name', NAMES)
def test_product_tables_and_state_dict_match_oracle(S, R, golden, name):
    g = golden[name]
    ref, _ = build_oracle_hyperprior(R, name)
    m = S.get_layer(name)
    assert type(m).__name__ == name and name in S.LAYER_CLASS_DICT
    assert sorted(m.state_dict().keys()) == sorted(ref.state_dict().keys())
    assert m.updated is False
    with pytest.raises(ValueError):
        m.gaussian_conditional._tables()                                # "Uninitialized CDFs. Run update() first"
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    assert m.update() is True and m.updated is True
    assert m.update() is False
    gc = m.gaussian_conditional
    assert hashlib.sha256(gc._quantized_cdf.numpy().tobytes()).hexdigest() == g['gc_cdf_sha256']
    assert torch.equal(gc._offset, g['gc_offset']) and torch.equal(gc._cdf_length, g['gc_cdf_length'])
    assert torch.equal(gc.scale_table, g['scale_table'])
    for i, row in g['gc_cdf_rows'].items():
        assert torch.equal(gc._quantized_cdf[i, :row.numel()], row)
    ref.update()
    assert torch.equal(m.entropy_bottleneck._quantized_cdf, ref.entropy_bottleneck._quantized_cdf)
    # a checkpoint saved after update() loads into a fresh module (buffers resized first, layer.py:706-720)
    fresh = S.get_layer(name)
    fresh.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert torch.equal(fresh.gaussian_conditional._quantized_cdf, gc._quantized_cdf)
    assert torch.equal(fresh.entropy_bottleneck._cdf_length, m.entropy_bottleneck._cdf_length)
    assert m.aux_loss().item() == pytest.approx(ref.aux_loss().item(), rel=1e-6)


def test_hyperprior_modes_of_the_oracle(R):
    """decode(encode(x)) equals the updated-train path of MSHP (both dequantise with the same means); the SHP
    updated-train path fails in the reference too: it expands the 16 hyper-latent medians against the 24-channel
    latent (layer.py:691-693)."""
    m, x = build_oracle_hyperprior(R, 'MSHPBasedResNetBottleneck')
    m.update()
    with torch.no_grad():
        a = m(x)
        m.train()
        b = m(x)
    assert torch.equal(a, b)
    s, x = build_oracle_hyperprior(R, 'SHPBasedResNetBottleneck')
    s.update()
    s.train()
    with torch.no_grad(), pytest.raises(RuntimeError):
        s(x)


def test_gaussian_conditional_argument_errors(S):
    with pytest.raises(ValueError):
        S.GaussianConditional([0.5, 0.1])          # not sorted
    with pytest.raises(ValueError):
        S.GaussianConditional([0.0, 1.0])          # non-positive entry
    with pytest.raises(ValueError):
        S.GaussianConditional(1.0)                 # wrong type
    with pytest.raises(ValueError):
        S.GaussianConditional(None, scale_bound=0.0)
    gc = S.GaussianConditional(None)
    with pytest.raises(ValueError):
        gc.quantize(torch.zeros(1, 2, 3), 'nearest')
    with pytest.raises(S.hip.Sc2Error):            # no CPU fallback
        gc.quantize(torch.zeros(1, 2, 3), 'symbols')
