"""-m gpu: the bottleneck module and the splittable ResNet through the reference's own API, against the
oracle, on the committed golden fixture and on seeded inputs at the BASELINE shape (224x224)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))


def _golden():
    return torch.load(os.path.join(HERE, 'golden', 'fp_golden.pt'), weights_only=False)


def _pair(S, R, dev):
    from recipe import build_oracle_bottleneck
    ref, x = build_oracle_bottleneck(R)
    m = S.FPBasedResNetBottleneck()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    return m, ref, x


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item(), (got - ref).abs().max().item()


# bf16 operands (8 bits of mantissa) through 5 layers: relative L2 error of a layer output vs the f32 oracle
BF16_REL_L2 = 1.5e-2


def test_analysis_synthesis_vs_oracle(S, R, dev):
    g = _golden()
    m, ref, x = _pair(S, R, dev)
    with torch.no_grad():
        latent = m.analysis(x.to(dev))
        r, mx = rel_err(latent, g['latent'])
        assert r < BF16_REL_L2, 'latent rel L2 {} max {}'.format(r, mx)
        dec = m.synthesis(g['y_hat_eval'].to(dev))
        r, mx = rel_err(dec, g['decoded'])
        assert r < BF16_REL_L2, 'decoder rel L2 {} max {}'.format(r, mx)
        # module-by-module path (nn.Sequential call, f32 NCHW between modules) == unfused NHWC pipeline, and the
        # conv+GDN fused launches (GDN applied to the f32 accumulators) stay within bf16 rounding of both
        seq = m.encoder(x.to(dev))
        m.fuse_gdn = False
        assert torch.equal(seq.cpu(), m.analysis(x.to(dev)).cpu())
        m.fuse_gdn = True
        r, _ = rel_err(latent, seq)
        assert r < 8e-3, 'fused vs unfused encoder rel L2 {}'.format(r)
        m.output_format = 'bf16_nhwc'
        dec_b = m.synthesis(g['y_hat_eval'].to(dev))
        assert dec_b.dtype == torch.bfloat16 and dec_b.shape == dec.shape
        assert dec_b.is_contiguous(memory_format=torch.channels_last)
        r, _ = rel_err(dec_b, dec)
        assert r < 6e-3


def test_modes_match_reference_semantics(S, R, dev):
    """layer.py:535-550: not updated -> encoder/EB/decoder; updated+train -> round+detach; updated+eval -> codec."""
    g = _golden()
    m, ref, x = _pair(S, R, dev)
    xd = x.to(dev)
    with torch.no_grad():
        out0 = m(xd)                      # not updated, eval: EB in dequantize mode
        m.update()
        assert m.updated
        out_codec = m(xd)                 # updated, eval: encode -> bytes -> decode
        m.train()
        out_train = m(xd)                 # updated, train: round(y - med) + med, detached
        m.eval()
    assert torch.equal(out0.cpu(), out_codec.cpu()), 'lossless coder must not change y_hat'
    assert torch.equal(out0.cpu(), out_train.cpu())
    # Two separate statements instead of one loose end-to-end bound:
    # (i) bf16 encoder: its symbol flips (a latent within bf16 rounding distance of a .5 boundary) are the encoder's, so the
    #     decoder is judged ON THE DEVICE'S BYTES through the oracle's decoder;
    ref.update(force=True)
    with torch.no_grad():
        enc = m.encode(xd)
        r, mx = rel_err(out0, ref.decode(**enc))
    assert r < 1.5e-2, 'decoder on the device bytes: rel L2 {} max {}'.format(r, mx)
    # (ii) reference-precision encoder: the symbols are the f32 oracle's (up to summation order), so the whole path is held
    #     against the pure-f32 golden output end to end
    m.set_encoder_precision('f32')
    with torch.no_grad():
        out32 = m(xd)
    m.set_encoder_precision('bf16')
    r, mx = rel_err(out32, g['decoded_from_strings'])
    assert r < 2e-2, 'f32 encoder, end to end vs the f32 golden output: rel L2 {} max {}'.format(r, mx)
    r_bf16, _ = rel_err(out0, g['decoded_from_strings'])
    assert r_bf16 < 0.1     # (the old bound, kept as a sanity fence only: it contains the bf16 encoder's symbol flips)


def test_encode_decode_bitstreams(S, R, dev):
    g = _golden()
    m, ref, x = _pair(S, R, dev)
    m.update()
    ref.update(force=True)
    eb, reb = m.entropy_bottleneck, ref.entropy_bottleneck
    assert torch.equal(eb._quantized_cdf.cpu(), g['quantized_cdf'])
    assert torch.equal(eb._offset.cpu(), g['offset']) and torch.equal(eb._cdf_length.cpu(), g['cdf_length'])
    with torch.no_grad():
        # (i) coder bit-exact given identical latents: the oracle's latent through the device coder
        strings = eb.compress(g['latent'].to(dev))
        assert [s.hex() for s in strings] == g['strings_hex']
        y_hat = eb.decompress(strings, g['shape'])
        assert torch.equal(y_hat.cpu(), g['y_hat_eval'])
        # (ii) the module API: encode() dict structure, pickled size, decode()
        enc = m.encode(x.to(dev))
        assert set(enc.keys()) == {'strings', 'shape'} and isinstance(enc['shape'], torch.Size)
        assert len(enc['strings']) == 1 and len(enc['strings'][0]) == x.shape[0]
        assert all(isinstance(s, bytes) and len(s) % 4 == 0 and len(s) >= 8 for s in enc['strings'][0])
        dev_latent = m.analysis(x.to(dev)).cpu()
        assert enc['strings'][0] == reb.compress(dev_latent), 'device coder != oracle coder on the device latent'
        sym_dev, sym_ref = reb.symbols(dev_latent), g['symbols']
        mismatch = (sym_dev != sym_ref).float().mean().item()
        assert mismatch < 0.02, 'symbol mismatch rate vs f32 oracle {}'.format(mismatch)
        a = S.FileSizeAnalyzer('KB')
        a.analyze(enc)
        assert a.file_size_list[0] == R.file_size(enc)
        dec = m.decode(**enc)
        ref_dec = ref.decode(**enc)
        r, mx = rel_err(dec, ref_dec)
        assert r < 1.5e-2, 'decode rel L2 {} max {}'.format(r, mx)
        # device-resident variant gives the same bytes without the host round trip
        buf, off, nb, st, shape = m.encode_device(x.to(dev))
        b, o, n = buf.cpu().numpy(), off.cpu().numpy(), nb.cpu().numpy()
        assert [b[i, o[i]:o[i] + n[i]].tobytes() for i in range(len(n))] == enc['strings'][0]
        assert torch.equal(m.decode_device(buf, off, nb, shape).cpu(), dec.cpu())


@pytest.mark.parametrize('N,H,W', [(1, 224, 224), (5, 224, 224), (2, 96, 160), (3, 33, 47)])
def test_roundtrip_properties_at_size(S, R, dev, N, H, W):
    """Size-independent properties at the BASELINE shape and on ragged shapes: lossless coder round trip,
    eval forward == codec forward, byte-identical streams vs the oracle coder on the device latent."""
    torch.manual_seed(N * 1000 + H)
    m, ref, _ = _pair(S, R, dev)
    m.update()
    ref.update(force=True)
    x = torch.rand(N, 3, H, W)
    with torch.no_grad():
        latent = m.analysis(x.to(dev))
        oh = ((H + 4 - 5) // 2 + 1 + 4 - 5) // 2 + 1 - 1
        ow = ((W + 4 - 5) // 2 + 1 + 4 - 5) // 2 + 1 - 1
        assert latent.shape == (N, 24, oh, ow)
        strings = m.entropy_bottleneck.compress(latent)
        assert strings == ref.entropy_bottleneck.compress(latent.cpu())
        y_hat = m.entropy_bottleneck.decompress(strings, latent.shape[-2:])
        y_q, _ = m.entropy_bottleneck(latent)
        assert torch.equal(y_hat, y_q)
        out = m(x.to(dev))
        assert out.shape == (N, 256, oh + 1, ow + 1)
        if N * H * W <= 5 * 224 * 224:
            ref_latent = ref.encoder(x)
            r, mx = rel_err(latent, ref_latent)
            assert r < 1.5e-2, 'latent rel L2 {} max {}'.format(r, mx)


def test_splittable_resnet_logits(S, R, dev):
    torch.manual_seed(0)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    model = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False, num_classes=1000,
                                analysis_config={'analyzes_after_compress': True,
                                                 'analyzer_configs': [{'key': 'FileSizeAnalyzer', 'kwargs': {'unit': 'KB'}}]})
    R.perturb_quantiles(model.bottleneck_layer.entropy_bottleneck)
    with torch.no_grad():
        model.bottleneck_layer.encoder[4].weight.mul_(40.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = R.SplittableResNet50(R.FPBasedResNetBottleneck())
    ref.load_state_dict(sd, strict=False)
    ref.bottleneck_layer.load_state_dict({k[len('bottleneck_layer.'):]: v for k, v in sd.items()
                                          if k.startswith('bottleneck_layer.')}, strict=False)
    model.eval().to(dev)
    ref.eval()
    x = torch.rand(2, 3, 64, 64)
    with torch.no_grad():
        ref_pre = ref(x)
        pre = model(x.to(dev))
        model.update()
        ref.update()
        model.activate_analysis()
        post = model(x.to(dev))
        ref_post = ref(x)
    scale = ref_pre.abs().max().item()
    assert (pre.cpu() - ref_pre).abs().max().item() <= 0.03 * scale + 0.03
    assert (post.cpu() - ref_post).abs().max().item() <= 0.03 * scale + 0.03
    assert len(model.analyzers[0].file_size_list) == 1 and model.analyzers[0].file_size_list[0] > 0
    # bf16 task head fed zero-copy by the decoder's NHWC output
    model.set_compute_dtype('bf16')
    with torch.no_grad():
        post_bf16, nb, st = model.forward_device(x.to(dev))
    assert int(st.max()) == 0 and nb.shape == (2,)
    assert model._hip_head is not None, 'bf16 eval must run the fused HIP head'
    # the bf16 decoder + head judged on the DEVICE'S bytes (the oracle decodes them and runs its f32 tail): no symbol flips in
    # the comparison, hence the tight bound; then the reference-precision encoder against the pure-f32 path end to end
    with torch.no_grad():
        enc = model.bottleneck_layer.encode(x.to(dev))
        h = ref.bottleneck_layer.decode(**enc)
        ref_on_bytes = ref.fc(torch.flatten(ref.avgpool(ref.layer4(ref.layer3(ref.layer2(h)))), 1))
        model.set_encoder_precision('f32')
        post_f32enc, _, _ = model.forward_device(x.to(dev))
        model.set_encoder_precision('bf16')
    assert (post_bf16.float().cpu() - ref_on_bytes).abs().max().item() <= 0.03 * scale + 0.03
    assert (post_f32enc.float().cpu() - ref_post).abs().max().item() <= 0.03 * scale + 0.03
    assert (post_bf16.float().cpu() - ref_post).abs().max().item() <= 0.08 * scale + 0.08   # (sanity fence: incl. symbol flips)
    # the same head through torch modules (MIOpen) agrees with the fused folded-BN head
    model.use_hip_head = False
    with torch.no_grad():
        post_torch, _, _ = model.forward_device(x.to(dev))
    assert (post_bf16.float() - post_torch.float()).abs().max().item() <= 0.05 * scale + 0.05


def test_hip_head_folded_bn(S, R, dev):
    """conv+BN(+ReLU)(+residual) fused launches vs the f32 torch modules with non-trivial BN statistics."""
    torch.manual_seed(3)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    model = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False, num_classes=1000)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    model.eval()
    x = torch.randn(3, 256, 14, 14)
    with torch.no_grad():
        ref = model.head(x)                       # f32 torch modules on CPU
    model.to(dev).set_compute_dtype('bf16')
    xb = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        out = model.head(xb)
    assert out.shape == (3, 1000) and out.dtype == torch.float32
    scale = ref.abs().max().item()
    assert (out.cpu() - ref).abs().max().item() <= 0.05 * scale + 0.05
