"""Host-side mirror of the reference interface: registries, module paths, state-dict keys, update(), losses,
data-size metric -- checked against the oracle and the committed golden fixture (CPU only)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from recipe import build_oracle_bottleneck, fingerprint  # noqa: E402


@pytest.fixture(scope='module')
def golden():
    return torch.load(os.path.join(HERE, 'golden', 'fp_golden.pt'), weights_only=False)


@pytest.fixture(scope='module')
def oracle_model(R, golden):
    m, x = build_oracle_bottleneck(R)
    fp = fingerprint(m)
    for k, v in golden['fingerprint'].items():
        assert abs(fp[k] - v) <= 1e-6 * max(1.0, abs(v)), 'seeded weights drifted for {}'.format(k)
    assert torch.equal(x, golden['x'])
    return m, x


def test_oracle_reproduces_golden(R, golden, oracle_model):
    m, x = oracle_model
    with torch.no_grad():
        latent = m.encoder(x)
        y_hat, lik = m.entropy_bottleneck(latent)
        torch.testing.assert_close(latent, golden['latent'], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lik, golden['lik_eval'], rtol=1e-4, atol=1e-7)
        yn, ln = m.entropy_bottleneck(golden['latent'], training=True, noise=golden['noise'])
        torch.testing.assert_close(yn, golden['y_hat_noise'], rtol=0, atol=1e-6)
        torch.testing.assert_close(ln, golden['lik_noise'], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(m.decoder(golden['y_hat_eval']), golden['decoded'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(m.encoder[3](golden['gdn_in']), golden['gdn_out'], rtol=1e-5, atol=1e-6)
        assert abs(float(R.bpp_loss(golden['y_hat_eval'], golden['lik_eval'], 'sum')) - float(golden['bits_eval'])) < 1e-2
    m.update(force=True)
    eb = m.entropy_bottleneck
    assert torch.equal(eb._quantized_cdf, golden['quantized_cdf'])
    assert torch.equal(eb._offset, golden['offset']) and torch.equal(eb._cdf_length, golden['cdf_length'])
    strings = eb.compress(golden['latent'])
    assert [s.hex() for s in strings] == golden['strings_hex']
    assert torch.equal(eb.symbols(golden['latent']), golden['symbols'])
    enc = {'strings': [strings], 'shape': torch.Size(golden['shape'])}
    assert R.file_size(enc) == golden['file_size_kb']
    dec = eb.decompress(strings, golden['shape'])
    torch.testing.assert_close(dec, golden['y_hat_eval'], rtol=0, atol=0)


def test_registry_and_state_dict_keys(S, R, oracle_model):
    assert S.LAYER_CLASS_DICT['FPBasedResNetBottleneck'] is S.FPBasedResNetBottleneck
    assert S.get_layer('nope') is None
    assert S.BACKBONE_FUNC_DICT['splittable_resnet'] is S.splittable_resnet
    assert 'splittable_resnet' in S.MODEL_DICT and 'FileSizeAnalyzer' in S.ANALYZER_CLASS_DICT
    m = S.get_layer('FPBasedResNetBottleneck', num_bottleneck_channels=24, num_target_channels=256)
    ref, _ = oracle_model
    assert sorted(m.state_dict().keys()) == sorted(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert m.state_dict()[k].shape == v.shape or k.startswith('entropy_bottleneck._'), k
    assert sum(p.numel() for p in m.parameters()) == 1304168
    assert not m.updated and m.entropy_bottleneck._offset.numel() == 0
    names = dict(m.named_modules())
    for path in ('encoder', 'decoder', 'entropy_bottleneck', 'encoder.1', 'decoder.3'):
        assert path in names


def test_update_tables_match_oracle_and_golden(S, golden, oracle_model):
    ref, _ = oracle_model
    m = S.FPBasedResNetBottleneck()
    tables = ('_offset', '_quantized_cdf', '_cdf_length')
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items() if not k.endswith(tables)}, strict=False)
    assert m.update() is True
    assert m.updated
    eb = m.entropy_bottleneck
    assert torch.equal(eb._quantized_cdf, golden['quantized_cdf'])
    assert torch.equal(eb._offset, golden['offset'])
    assert torch.equal(eb._cdf_length, golden['cdf_length'])
    assert m.update() is False and m.update(force=True) is True
    torch.testing.assert_close(m.aux_loss(), golden['aux_loss'])
    # checkpoint with updated tables reloads into a fresh module (buffers are resized first)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = S.FPBasedResNetBottleneck()
    m2.load_state_dict(sd)
    assert torch.equal(m2.entropy_bottleneck._quantized_cdf, golden['quantized_cdf'])
    # compressai <= 1.1 key names are remapped
    old = {}
    for k, v in sd.items():
        for new, o in (('matrices.', '_matrix'), ('biases.', '_bias'), ('factors.', '_factor')):
            k = k.replace('entropy_bottleneck.' + new, 'entropy_bottleneck.' + o)
        old[k] = v
    assert any('_matrix0' in k for k in old)
    m3 = S.FPBasedResNetBottleneck()
    m3.load_state_dict(old)
    assert torch.equal(m3.entropy_bottleneck.matrices[2], m.entropy_bottleneck.matrices[2])


def test_uninitialised_tables_raise(S):
    m = S.FPBasedResNetBottleneck()
    with pytest.raises(ValueError, match='update'):
        m.entropy_bottleneck._tables()
    with pytest.raises(ValueError, match='Invalid quantization mode'):
        m.entropy_bottleneck.quantize(torch.zeros(1, 24, 2, 2), 'bogus')


def test_splittable_resnet_contract(S):
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    m = S.splittable_resnet(cfg, resnet_name='resnet50', skips_avgpool=False, skips_fc=False, num_classes=1000,
                            analysis_config={'analyzes_after_compress': True,
                                             'analyzer_configs': [{'key': 'FileSizeAnalyzer', 'kwargs': {'unit': 'KB'}}]})
    assert S.check_if_updatable(m) and m.get_aux_module() is m.bottleneck_layer
    assert [n for n, _ in m.named_children()] == ['bottleneck_layer', 'layer2', 'layer3', 'layer4', 'avgpool', 'fc']
    keys = list(m.state_dict().keys())
    assert 'layer2.0.conv1.weight' in keys and 'layer4.2.bn3.running_var' in keys and 'fc.bias' in keys
    assert 'bottleneck_layer.entropy_bottleneck.quantiles' in keys
    m.update()
    assert m.bottleneck_updated and m.bottleneck_layer.updated
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False)
    m2.load_state_dict(sd)
    assert not any(k.startswith('bottleneck_layer.') for k in sd)  # popped, as the reference does
    assert torch.equal(m2.bottleneck_layer.entropy_bottleneck._quantized_cdf,
                       m.bottleneck_layer.entropy_bottleneck._quantized_cdf)
    with pytest.raises(KeyError):
        S.splittable_resnet(cfg, resnet_name='resnet18')


def test_bpp_loss_and_file_size(S, R, golden):
    io = {'bottleneck_layer.entropy_bottleneck': {'output': (golden['y_hat_eval'], golden['lik_eval'])}}
    for red, key in (('sum', 'bits_eval'), ('mean', 'bpp_mean'), ('batchmean', 'bpp_batchmean')):
        got = S.BppLoss('bottleneck_layer.entropy_bottleneck', reduction=red)(io)
        torch.testing.assert_close(got, golden[key])
    enc = {'strings': [[bytes.fromhex(h) for h in golden['strings_hex']]], 'shape': torch.Size(golden['shape'])}
    a = S.FileSizeAnalyzer(unit='KB')
    a.analyze(enc)
    assert a.file_size_list == [golden['file_size_kb']] == [R.file_size(enc)]
    b = S.FileSizeAnalyzer(unit='B')
    b.analyze({'strings': [[b'\x00' * 8]], 'shape': torch.Size([1, 1])})
    assert b.file_size_list[0] == R.file_size({'strings': [[b'\x00' * 8]], 'shape': torch.Size([1, 1])}, 1)
    acc = S.FileSizeAccumulator(unit='KB')
    acc.analyze(2048)
    assert acc.file_size_list == [2.0] and acc.summary()['count'] == 1


def test_pack_conv_weight_gather_equals_the_layout_chain(S):
    """hip.pack_conv_weight packs through a cached gather map (one cast + one index_select); the chain of layout ops it replaced stays as
    `_pack_conv_weight_reference`: the two are bit-identical for every k order / tile / fragment layout, for padded and unpadded
    shapes, and for the data gradient's sub-filters packed straight from the parameter (`_sub`)."""
    import torch
    hip = S.hip
    torch.manual_seed(0)
    for shape in [(96, 3, 5, 5), (48, 96, 5, 5), (24, 48, 2, 2), (512, 24, 2, 2), (256, 512, 2, 2), (128, 128, 3, 3), (1024, 256, 1, 1),
                  (40, 72, 3, 3), (8, 8, 3, 3), (136, 264, 2, 2)]:
        w = torch.randn(shape)
        for order in (hip.K_TAP_MAJOR, hip.K_SLAB_MAJOR, hip.K_TAP_MAJOR | hip.K_B_TILE_MAJOR, hip.K_SLAB_MAJOR | hip.K_B_TILE_MAJOR,
                      hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR, hip.K_TAP_MAJOR | hip.K_B_FRAG_MAJOR):
            if (order & 1) == hip.K_SLAB_MAJOR and shape[1] % 32:
                continue
            if (order & hip.K_B_FRAG_MAJOR) and hip.weight_rows(shape[0]) % 16:
                continue
            a, b = hip.pack_conv_weight(w, order), hip._pack_conv_weight_reference(w, order)
            assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), (shape, order)
        wt = w.permute(1, 0, 2, 3)
        for st, pd in ((1, 0), (1, 1), (2, 1), (2, 2)):
            for ch in range(st):
                for cw in range(st):
                    rh, rw = (ch + pd) % st, (cw + pd) % st
                    if rh >= shape[2] or rw >= shape[3]:
                        continue
                    sub = wt[:, :, rh::st, rw::st].flip(2, 3).contiguous()
                    a = hip.pack_conv_weight(w, hip.K_TAP_MAJOR, _sub=(rh, st, rw, st))
                    b = hip._pack_conv_weight_reference(sub)
                    assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), (shape, st, pd, ch, cw)



def test_mse_sink_entries_belong_to_one_backward_pass(S):
    """autograd.MseSink: a consumer hands (operands) to its producer through the sink instead of returning a gradient.
    An entry written in a backward pass in which the producer's backward never runs must not be applied by a later pass, and a
    retained graph run twice applies each pass's entry once -- on CPU stand-ins with the very hand-over protocol of
    frozen.MseSumFn / FrozenStackFn."""
    from sc2bench_amd.autograd import MseSink
    applied = []

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.sink = MseSink()
            ctx.set_materialize_grads(False)
            y = x * 2.0
            y._sink = ctx.sink
            return y

        @staticmethod
        def backward(ctx, gy):
            entries = ctx.sink.drain()
            applied.append(len(entries))
            g = gy if gy is not None else 0.0
            for (t,) in entries:
                g = g + t
            return g * 2.0

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y, sink):
            ctx.sink = sink
            return y.sum()

        @staticmethod
        def backward(ctx, g):
            ctx.sink.put((torch.ones(3) * g,))
            return None, None

    x = torch.zeros(3, requires_grad=True)
    y = Producer.apply(x)
    loss = Consumer.apply(y, y._sink)
    # pass 1 stops at the feature: the producer's backward does not run, the entry stays behind (and the tensor-level gradient
    # is None -- the documented restriction)
    assert torch.autograd.grad(loss, inputs=[y], retain_graph=True, allow_unused=True)[0] is None
    assert len(y._sink) == 1 and applied == []
    # pass 2 runs the whole graph: only ITS entry is applied
    loss.backward(retain_graph=True)
    assert applied == [1] and torch.equal(x.grad, torch.full((3,), 2.0)) and len(y._sink) == 0
    # pass 3 on the retained graph: again exactly one
    loss.backward()
    assert applied == [1, 1] and torch.equal(x.grad, torch.full((3,), 4.0))


def test_host_steps_follow_the_measured_host_speed():
    """StagePipeline.host_steps_from_measurement: the number of leading batches given to the host coder is bounded by what warm() measured
    on this host -- a batch leaves the host path after about (k + 1) x the measured time, and is worth taking only while that is earlier
    than the device coder's first result (~21 ms)."""
    from sc2bench_amd.pipeline import StagePipeline as P
    assert P.host_steps_from_measurement(None) == 4 and P.host_steps_from_measurement(0.0) == 4      # nothing measured: the model's answer stands
    assert P.host_steps_from_measurement(6.2) == 4 and P.host_steps_from_measurement(8.0) == 4      # the boxes of the round: 6 - 8 ms end to end
    assert P.host_steps_from_measurement(10.0) == 3
    assert P.host_steps_from_measurement(14.0) == 2
    assert P.host_steps_from_measurement(20.0) == 1
    assert P.host_steps_from_measurement(40.0) == 0       # a host that busy takes none
    assert P.host_steps_from_measurement(6.2, limit=2) == 2
