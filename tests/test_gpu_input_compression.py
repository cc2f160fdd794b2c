"""-m gpu: SURVEY.md 8(f) rank 3 -- factorized-prior neural INPUT compression (BASELINE config 3) on the HIP library:
squared-form GDN, biased k5 s2 convolutions / transposed convolutions (output_padding 1) at N = 192 / M = 320, the
320-channel entropy bottleneck through the batched range coder, `AdaptivePad(64)`, `NeuralInputCompressionClassifier`;
against the oracle (oracle/cpu_ref_input.py) and the committed fixture."""
import hashlib
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
BF16_REL_L2 = 1.5e-2      # bf16 operands through a stack of layers vs the f32 oracle (as tests/test_gpu_bottleneck.py)


def _golden():
    return torch.load(os.path.join(HERE, 'golden', 'input_golden.pt'), weights_only=False)


def rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item()


def _pair(S, R, dev):
    from oracle import cpu_ref_input as RI
    from recipe import build_oracle_factorized_prior
    ref, x = build_oracle_factorized_prior(RI, R)
    m = S.bmshj2018_factorized(quality=8)
    assert isinstance(m, S.FactorizedPrior) and (m.N, m.M) == (192, 320)
    assert set(m.state_dict().keys()) == set(ref.state_dict().keys())
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    return m.eval().to(dev), ref, x


def test_gdn_squared_form_kernel(S, R, dev):
    g = _golden()
    m, ref, _ = _pair(S, R, dev)
    x = g['gdn_in']
    xb = x.to(torch.bfloat16).float()                       # the kernel's operand precision
    with torch.no_grad():
        for mod, rmod, key in ((m.g_a[1], ref.g_a[1], 'gdn_out'), (m.g_s[1], ref.g_s[1], 'igdn_out')):
            out = mod(x.to(dev))
            assert out.shape == x.shape and out.dtype == torch.float32
            want = rmod(xb)
            # bf16 operands (x, x^2, gamma), f32 accumulation: 2^-8 relative per element + the operand rounding
            err = (out.cpu() - want).abs()
            assert float((err / (want.abs() + 2e-2)).max()) < 2.5e-2, key
            assert rel(out, g[key]) < 8e-3, key
        # odd spatial sizes and a batch that is not a tile multiple
        x2 = torch.randn(3, 192, 9, 13)
        assert rel(m.g_a[3](x2.to(dev)), ref.g_a[3](x2)) < 8e-3


def test_biased_conv_and_deconv_layers(S, R, dev):
    m, ref, _ = _pair(S, R, dev)
    torch.manual_seed(4)
    with torch.no_grad():
        x = torch.rand(2, 3, 64, 96)
        assert rel(m.g_a[0](x.to(dev)), ref.g_a[0](x)) < 6e-3          # 3 -> 192, k5 s2 p2, bias
        h = torch.randn(2, 192, 16, 24)
        assert rel(m.g_a[2](h.to(dev)), ref.g_a[2](h)) < 6e-3          # 192 -> 192
        assert rel(m.g_a[6](h.to(dev)), ref.g_a[6](h)) < 6e-3          # 192 -> 320
        y = torch.randn(2, 320, 4, 6)
        d = m.g_s[0](y.to(dev))
        assert d.shape == (2, 192, 8, 12)                               # (H-1)*2 - 4 + 5 + 1 = 2H
        assert rel(d, ref.g_s[0](y)) < 6e-3                             # transposed conv, output_padding 1, bias
        h2 = torch.randn(1, 192, 7, 5)
        assert rel(m.g_s[2](h2.to(dev)), ref.g_s[2](h2)) < 6e-3
        out = m.g_s[6](h.to(dev))                                       # 192 -> 3 (output channels padded to 8 inside)
        assert out.shape == (2, 3, 32, 48) and rel(out, ref.g_s[6](h)) < 6e-3


def test_factorized_prior_codec_vs_oracle(S, R, dev):
    g = _golden()
    m, ref, x = _pair(S, R, dev)
    m.update()
    ref.update(force=True)
    eb, reb = m.entropy_bottleneck, ref.entropy_bottleneck
    assert hashlib.sha256(eb._quantized_cdf.cpu().numpy().tobytes()).hexdigest() == g['cdf_sha256']
    assert torch.equal(eb._offset.cpu(), g['offset']) and torch.equal(eb._cdf_length.cpu(), g['cdf_length'])
    with torch.no_grad():
        y = m.analysis(x.to(dev))
        assert y.shape == (2, 320, 4, 8) and rel(y, g['y']) < BF16_REL_L2
        # integer half, bit-exact: the oracle's latent through the device coder == the fixture's streams
        strings = eb.compress(g['y'].to(dev))
        assert [s.hex() for s in strings] == g['strings_hex']
        assert torch.equal(eb.decompress(strings, g['shape']).cpu(), reb.decompress(strings, g['shape']))
        # module API on the device latent: bytes equal the oracle coder's on the same latent
        obj = m.compress(x.to(dev))
        assert set(obj) == {'strings', 'shape'} and tuple(obj['shape']) == (4, 8) and len(obj['strings'][0]) == 2
        assert obj['strings'][0] == reb.compress(y.cpu())
        sym_dev, sym_ref = reb.symbols(y.cpu()), reb.symbols(g['y'])
        assert (sym_dev != sym_ref).float().mean().item() < 0.03
        out = m.decompress(**obj)
        assert set(out) == {'x_hat'} and out['x_hat'].shape == (2, 3, 64, 128)
        assert float(out['x_hat'].min()) >= 0.0 and float(out['x_hat'].max()) <= 1.0
        ref_x_hat = ref.decompress(**obj)['x_hat']                      # same bytes through the oracle's g_s
        assert rel(out['x_hat'], ref_x_hat) < BF16_REL_L2
        assert rel(m.synthesis(g['y'].round().to(dev)), ref.g_s(g['y'].round())) < BF16_REL_L2
        fwd = m(x.to(dev))
        assert set(fwd) == {'x_hat', 'likelihoods'} and fwd['likelihoods']['y'].shape == y.shape
        a = S.FileSizeAnalyzer('KB')
        a.analyze(obj)
        assert a.file_size_list[0] == R.file_size(obj)


def test_neural_input_compression_classifier_config3(S, R, dev):
    """The wrapper as BASELINE config 3 builds it: AdaptivePad(64) -> compress -> size -> decompress -> CenterCrop +
    Normalize -> ResNet-50, at 224 x 224, batch 1 and batch 4."""
    from oracle import cpu_ref_input as RI
    from sc2bench_amd import transforms as T
    from sc2bench_amd.resnet import resnet50
    m, ref, _ = _pair(S, R, dev)
    m.update()
    ref.update(force=True)
    torch.manual_seed(5)
    clf = resnet50(num_classes=1000).eval()
    post = T.Compose([T.CenterCrop([224, 224]), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    wrapped = S.NeuralInputCompressionClassifier(
        clf, pre_transform=T.AdaptivePad(fill=0, factor=64), compression_model=m, post_transform=post,
        analysis_config={'analyzes_after_compress': True, 'analyzer_configs': [{'key': 'FileSizeAnalyzer', 'kwargs': {'unit': 'KB'}}]})
    wrapped.eval().to(dev)
    wrapped.activate_analysis()
    x = torch.rand(4, 3, 224, 224)
    with torch.no_grad():
        for xb in (x[:1], x):
            out = wrapped(xb.to(dev))
            padded = RI.adaptive_pad(xb, factor=64)
            obj = m.compress(padded.to(dev))                           # the device's bytes, decoded by the oracle
            ref_out = post(ref.decompress(**obj)['x_hat'])
            ref_logits = clf.cpu()(ref_out)
            clf.to(dev)
            assert out.shape == (xb.shape[0], 1000)
            scale = ref_logits.abs().max().item()
            assert (out.float().cpu() - ref_logits).abs().max().item() <= 0.03 * scale + 0.03
    sizes = wrapped.analyzers[0].file_size_list
    assert len(sizes) == 2 and all(s > 0 for s in sizes)        # one analysed object per forward call
    assert sizes[1] == R.file_size(obj)                          # the batch-of-4 object, pickled as the reference pickles it
    # round 5: set_compute_dtype('bf16') -- the ResNet-50 classifier on the library's fused conv + norm kernels (head.HipResNet:
    # stem conv 7x7 + layer1..4 + fc); against the f32 torch classifier on the same reconstruction, at the bf16 head's tolerance
    with torch.no_grad():
        f32_logits = wrapped(x.to(dev)).float().cpu()
        wrapped.set_compute_dtype('bf16')
        bf_logits = wrapped(x.to(dev)).float().cpu()
    assert wrapped.__dict__.get('_hip_clf') is not None, 'the bf16 mode did not take the HIP classifier'
    scale = f32_logits.abs().max().item()
    assert (bf_logits - f32_logits).abs().max().item() <= 0.03 * scale + 0.03
