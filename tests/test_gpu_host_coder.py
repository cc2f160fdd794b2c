"""-m gpu: the two coders of the product give the same bytes and the same decoded values -- the library's HOST coder
(csrc/rans_host.cpp, used for up to SC2_HOST_CODER_MAX_STREAMS streams: the reference's evaluation mode codes one per
forward, script/task/image_classification.py:106-145) and the batched DEVICE coder (csrc/rans.hip) -- through the reference's
API (`compress` / `decompress`, `encode` / `decode`), and both equal the oracle's coder."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))


def _model(S, R, dev):
    from recipe import build_oracle_bottleneck
    ref, x = build_oracle_bottleneck(R)
    m = S.FPBasedResNetBottleneck()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    m.update()
    ref.update(force=True)
    return m, ref


@pytest.mark.parametrize('n', [1, 3, 8, 9])
def test_host_and_device_coder_agree(S, R, dev, monkeypatch, n):
    m, ref = _model(S, R, dev)
    eb = m.entropy_bottleneck
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, 64, 96, generator=g).to(dev)
    with torch.no_grad():
        latent = m.analysis(x)
        latent[0, 0, 0, 0] = 5000.0        # escapes of both signs
        latent[-1, 5, 1, 2] = -321.0
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '0')
        dev_strings = eb.compress(latent)
        dev_y = eb.decompress(dev_strings, latent.shape[-2:])
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '8')
        assert (S.hip.host_coder_max_streams() >= n) == (n <= 8)
        strings = eb.compress(latent)
        y = eb.decompress(strings, latent.shape[-2:])
    assert strings == dev_strings
    assert strings == ref.entropy_bottleneck.compress(latent.cpu())
    assert torch.equal(y.cpu(), dev_y.cpu())
    assert torch.equal(y.cpu(), ref.entropy_bottleneck.decompress(strings, latent.shape[-2:]))


def test_encode_decode_module_api_bs1(S, R, dev, monkeypatch):
    """encode() / decode() at the reference's evaluation batch size: same dict, same bytes, same decoder output whichever
    coder runs; a table change (update(force=True) after new quantiles) reaches the host coder's prepared tables."""
    m, ref = _model(S, R, dev)
    x = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        enc = m.encode(x)
        out = m.decode(**enc)
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '0')
        enc_d = m.encode(x)
        out_d = m.decode(**enc_d)
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '8')
        assert enc['strings'] == enc_d['strings'] and enc['shape'] == enc_d['shape'] == torch.Size([55, 55])
        assert isinstance(enc['strings'][0][0], bytes)
        assert torch.equal(out.cpu(), out_d.cpu())
        assert enc['strings'][0] == ref.entropy_bottleneck.compress(m.analysis(x).cpu())
        a = S.FileSizeAnalyzer('KB')
        a.analyze(enc)
        assert a.file_size_list[0] == R.file_size(enc)
        # new tables
        m.entropy_bottleneck.quantiles.data[:, 0, 0] -= 2.0
        m.entropy_bottleneck.quantiles.data[:, 0, 2] += 1.0
        m.update(force=True)
        enc2 = m.encode(x)
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '0')
        assert m.encode(x)['strings'] == enc2['strings'] and enc2['strings'] != enc['strings']


@pytest.mark.parametrize('name', ['SHPBasedResNetBottleneck', 'MSHPBasedResNetBottleneck'])
def test_hyperprior_streams_either_coder(S, R, dev, monkeypatch, name):
    from recipe import build_oracle_hyperprior
    ref, x = build_oracle_hyperprior(R, name)
    m = getattr(S, name)()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    m.update()
    with torch.no_grad():
        enc = m.encode(x.to(dev))
        out = m.decode(**enc)
        monkeypatch.setenv('SC2_HOST_CODER_MAX_STREAMS', '0')
        enc_d = m.encode(x.to(dev))
        out_d = m.decode(**enc_d)
    assert enc['strings'] == enc_d['strings']
    assert torch.equal(out.cpu(), out_d.cpu())
