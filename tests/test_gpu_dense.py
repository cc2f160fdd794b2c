"""-m gpu: the detection / segmentation callers driving the HIP bottleneck (BASELINE configs 4 and 5): the
feature-extraction body (FrozenBatchNorm2d backbone + FPN; dilated backbone + DeepLabv3 head) in f32 and in bf16 (HIP
ResNet stacks), before and after update(), against the oracle bottleneck composed with the same torch tail modules."""
import copy
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_features(R, model_body, ref_bn, x, keys, updated):
    """The oracle's bottleneck (decode(encode(x)) once updated), then CPU f32 copies of the body's tail modules."""
    with torch.no_grad():
        h = ref_bn.decode(**ref_bn.encode(x)) if updated else ref_bn(x)
        out = OrderedDict()
        for name, module in model_body.named_children():
            if name != 'bottleneck_layer':
                h = copy.deepcopy(module).cpu().float()(h)
            if name in keys:
                out[keys[name]] = h
    return out


def _build(S, R, dev, **resnet_kwargs):
    torch.manual_seed(7)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    backbone = S.splittable_resnet(cfg, skips_avgpool=True, skips_fc=True, **resnet_kwargs)
    R.perturb_quantiles(backbone.bottleneck_layer.entropy_bottleneck)
    with torch.no_grad():
        backbone.bottleneck_layer.encoder[4].weight.mul_(30.0)
        for m in backbone.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, S.resnet.FrozenBatchNorm2d)):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    ref_bn = R.FPBasedResNetBottleneck()
    ref_bn.load_state_dict({k: v.clone() for k, v in backbone.bottleneck_layer.state_dict().items()}, strict=False)
    ref_bn.eval()
    return backbone, ref_bn


def _close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape
    r = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    assert r < tol, 'rel L2 {}'.format(r)


def test_backbone_with_fpn_frozen_bn(S, R, dev):
    from sc2bench_amd import dense
    backbone, ref_bn = _build(S, R, dev, norm_layer='FrozenBatchNorm2d')
    keys = {'bottleneck_layer': '1', 'layer2': '2', 'layer3': '3', 'layer4': '4'}
    bf = dense.backbone_with_fpn(backbone, return_layer_dict=keys, in_channels_list=[256, 512, 1024, 2048], out_channels=256,
                                 analyzable_layer_key='bottleneck_layer',
                                 analysis_config={'analyzes_after_compress': True,
                                                  'analyzer_configs': [{'key': 'FileSizeAnalyzer', 'kwargs': {'unit': 'KB'}}]})
    assert isinstance(bf.body.layer2[0].bn1, S.resnet.FrozenBatchNorm2d)
    bf.eval().to(dev)
    x = torch.rand(1, 3, 320, 416)                  # a (small) variable-size detection input: H, W multiples of 32
    with torch.no_grad():
        feats = bf.body(x.to(dev))
        ref = _oracle_features(R, bf.body, ref_bn, x, keys, updated=False)
        assert [tuple(v.shape) for v in feats.values()] == [(1, 256, 80, 104), (1, 512, 40, 52), (1, 1024, 20, 26), (1, 2048, 10, 13)]
        for k in ref:      # behind the quantiser: a bf16 latent within rounding distance of .5 flips a symbol (inherent)
            _close(feats[k], ref[k], 0.12)
        pyramid = bf(x.to(dev))
        assert list(pyramid.keys()) == ['1', '2', '3', '4', 'pool'] and pyramid['pool'].shape == (1, 256, 5, 7)
        # updated: encode -> size -> decode inside the body; bf16: the undilated stacks on the HIP head
        bf.update()
        ref_bn.update(force=True)
        bf.activate_analysis()
        feats_u = bf.body(x.to(dev))
        ref_u = _oracle_features(R, bf.body, ref_bn, x, keys, updated=True)
        for k in ref_u:
            _close(feats_u[k], ref_u[k], 0.12)     # symbols may flip where bf16 moves a latent across a rounding boundary
        assert len(bf.body.analyzers[0].file_size_list) == 1 and bf.body.analyzers[0].file_size_list[0] > 0
        bf.body.set_compute_dtype('bf16')
        feats_b = bf.body(x.to(dev))
        assert feats_b['4'].dtype == torch.bfloat16 and set(bf.body._hip_layers) == {'layer2', 'layer3', 'layer4'}
        for k in feats_u:
            _close(feats_b[k], feats_u[k], 3e-2)
        pyr_b = bf(x.to(dev))
        _close(pyr_b['1'], bf.fpn(OrderedDict((k, v.float()) for k, v in feats_b.items()))['1'].cpu(), 1e-3)


def test_deeplabv3_dilated_backbone(S, R, dev):
    from sc2bench_amd import dense
    backbone, ref_bn = _build(S, R, dev, replace_stride_with_dilation=[False, True, True])
    keys = {'layer3': 'aux', 'layer4': 'out'}
    body = S.FeatureExtractionBackbone(backbone, keys, [{'key': 'FileSizeAnalyzer', 'kwargs': {'unit': 'KB'}}], True,
                                       analyzable_layer_key='bottleneck_layer')
    model = dense.create_deeplabv3(body, num_input_channels=2048, uses_aux=True, num_aux_channels=1024, num_classes=21)
    assert body.layer3[1].conv2.dilation == (2, 2) and body.layer4[1].conv2.dilation == (4, 4)
    model.eval().to(dev)
    x = torch.rand(1, 3, 257, 257)
    with torch.no_grad():
        feats = body(x.to(dev))
        ref = _oracle_features(R, body, ref_bn, x, keys, updated=False)
        assert feats['out'].shape == (1, 2048, 33, 33) and feats['aux'].shape == (1, 1024, 33, 33)
        for k in ref:
            _close(feats[k], ref[k], 0.12)
        out = model(x.to(dev))
        cls_ref = torch.nn.functional.interpolate(copy.deepcopy(model.classifier).cpu()(ref['out']), size=(257, 257),
                                                  mode='bilinear', align_corners=False)
        _close(out['out'], cls_ref, 0.15)
        # the VOC shape of the config (513 x 513, odd width), updated, bf16: layer2 AND the dilated stacks on the HIP head
        model.update()
        model.activate_analysis()
        body.set_compute_dtype('bf16')
        model.classifier.to(torch.bfloat16)
        model.aux_classifier.to(torch.bfloat16)
        x2 = torch.rand(2, 3, 513, 513)
        out2 = model(x2.to(dev))
        assert out2['out'].shape == (2, 21, 513, 513) and out2['aux'].shape == (2, 21, 513, 513)
        assert torch.isfinite(out2['out'].float()).all()
        assert set(body._hip_layers) == {'layer2', 'layer3', 'layer4'}
        assert len(body.analyzers[0].file_size_list) == 1
        ev = S.SegEvaluator(21)
        ev.update(torch.randint(0, 21, (2, 513, 513), device=dev).flatten(), out2['out'].argmax(1).flatten())
        assert int(ev.mat.sum()) == 2 * 513 * 513


@pytest.mark.parametrize('d,H,W', [(2, 33, 33), (4, 33, 35), (2, 5, 3), (4, 3, 2)])
def test_dilated_conv3x3_on_phase_grids(S, dev, d, H, W):
    """head._Conv on a dilated 3x3 layer (dilation d, padding d, stride 1: DeepLab's layer3 / layer4) = d * d undilated
    launches on the phase grids x[a::d, b::d]; against torch's dilated conv + folded BN + ReLU on the bf16-rounded operands
    (maps smaller than the dilation included)."""
    from sc2bench_amd import head, hip
    torch.manual_seed(d * 100 + H)
    conv = torch.nn.Conv2d(64, 128, 3, padding=d, dilation=d, bias=False)
    bn = torch.nn.BatchNorm2d(128).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.1)
    x = torch.randn(2, 64, H, W)
    c = head._Conv(conv.to(dev), bn.to(dev), 'test.dilated')
    with torch.no_grad():
        out = c(hip.nchw_f32_to_nhwc_bf16(x.to(dev)), hip.EPI_BIAS_RELU)
        w_f, b_f = head._fold(conv, bn)[3].cpu(), head._fold(conv, bn)[1].cpu()
        xr = x.to(torch.bfloat16).float()
        ref = torch.relu(torch.nn.functional.conv2d(xr, w_f.to(torch.bfloat16).float(), padding=d, dilation=d) + b_f.view(1, -1, 1, 1))
    assert out.shape == (2, H, W, 128)
    _close(out.permute(0, 3, 1, 2).float(), ref, 2e-2)


def _oracle_features_from_bytes(model_body, ref_bn, enc, keys):
    """The DEVICE's byte streams through the oracle's decoder, then CPU f32 copies of the body's tail modules: isolates the
    decoder + tail arithmetic from symbol flips of the bf16 encoder (the construction of tests/test_gpu_shapes.py)."""
    with torch.no_grad():
        h = ref_bn.decode(**enc)
        out = OrderedDict()
        for name, module in model_body.named_children():
            if name != 'bottleneck_layer':
                h = copy.deepcopy(module).cpu().float()(h)
            if name in keys:
                out[keys[name]] = h
    return out


def test_fpn_body_at_800x1216_bf16_vs_oracle_tail(S, R, dev):
    """BASELINE config 4's operating shape (a typical Faster R-CNN batch element, 800 x 1216): the FrozenBatchNorm2d body in
    bf16 -- bottleneck through encode / decode, layer2..4 on the HIP stacks at 100 x 152 / 50 x 76 / 25 x 38 -- against the
    oracle's decoder + f32 tail ON THE DEVICE'S BYTES, 3e-2 relative L2 per returned map (sc2bench/models/backbone.py:90-172,
    models/detection/base.py:44-129)."""
    from sc2bench_amd import dense
    torch.set_num_threads(32)
    backbone, ref_bn = _build(S, R, dev, norm_layer='FrozenBatchNorm2d')
    keys = {'bottleneck_layer': '1', 'layer2': '2', 'layer3': '3', 'layer4': '4'}
    bf = dense.backbone_with_fpn(backbone, return_layer_dict=keys, in_channels_list=[256, 512, 1024, 2048], out_channels=256,
                                 analyzable_layer_key='bottleneck_layer', analysis_config={'analyzes_after_compress': False})
    bf.eval().to(dev)
    bf.update()
    ref_bn.update(force=True)
    bf.body.set_compute_dtype('bf16')
    x = torch.rand(1, 3, 800, 1216, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        feats = bf.body(x.to(dev))
        assert set(bf.body._hip_layers) == {'layer2', 'layer3', 'layer4'}
        assert [tuple(v.shape) for v in feats.values()] == [(1, 256, 200, 304), (1, 512, 100, 152), (1, 1024, 50, 76),
                                                            (1, 2048, 25, 38)]
        enc = bf.body.bottleneck_layer.encode(x.to(dev))
        ref = _oracle_features_from_bytes(bf.body, ref_bn, enc, keys)
        for k in ref:
            _close(feats[k], ref[k], 3e-2)
        pyramid = bf(x.to(dev))
        assert pyramid['pool'].shape == (1, 256, 13, 19) and all(torch.isfinite(v.float()).all() for v in pyramid.values())


def test_deeplab_body_at_513_bf16_vs_oracle_tail(S, R, dev):
    """BASELINE config 5's shape (2 x 513 x 513): the dilated body in bf16 (layer2 and the dilated layer3 / layer4 on the HIP
    head) and the DeepLabv3 classifier against the oracle's decoder + f32 tail + f32 classifier on the device's bytes
    (sc2bench/models/segmentation/deeplabv3.py:44-104)."""
    from sc2bench_amd import dense
    torch.set_num_threads(32)
    backbone, ref_bn = _build(S, R, dev, replace_stride_with_dilation=[False, True, True])
    keys = {'layer3': 'aux', 'layer4': 'out'}
    body = S.FeatureExtractionBackbone(backbone, keys, [], False, analyzable_layer_key='bottleneck_layer')
    model = dense.create_deeplabv3(body, num_input_channels=2048, uses_aux=True, num_aux_channels=1024, num_classes=21)
    model.eval().to(dev)
    model.update()
    ref_bn.update(force=True)
    cls_ref = copy.deepcopy(model.classifier).cpu().float()
    body.set_compute_dtype('bf16')
    model.classifier.to(torch.bfloat16)
    model.aux_classifier.to(torch.bfloat16)
    x = torch.rand(2, 3, 513, 513, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        feats = body(x.to(dev))
        assert set(body._hip_layers) == {'layer2', 'layer3', 'layer4'}
        assert feats['out'].shape == (2, 2048, 65, 65) and feats['aux'].shape == (2, 1024, 65, 65)
        enc = body.bottleneck_layer.encode(x.to(dev))
        ref = _oracle_features_from_bytes(body, ref_bn, enc, keys)
        for k in ref:
            _close(feats[k], ref[k], 3e-2)
        out = model(x.to(dev))
        want = torch.nn.functional.interpolate(cls_ref(ref['out']), size=(513, 513), mode='bilinear', align_corners=False)
        _close(out['out'], want, 5e-2)      # + a bf16 ASPP head (torch ops) on top of the 3e-2 features


def _randomise_norms(module):
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
                m.bias.normal_(0, 0.1)


@pytest.mark.parametrize('N,H,W', [(2, 65, 65), (1, 33, 41)])
def test_hip_dense_head_deeplab_and_fcn_vs_torch_f32(S, dev, N, H, W):
    """Round 5: DeepLabHead (ASPP: 1x1, three ATROUS 3x3 convs at rates 12 / 24 / 36 through the descriptor's dilation field,
    the pooling branch, projection; then 3x3 + classifier) and FCNHead folded onto the library's kernels, against the same
    torch modules in f32 on the same bf16-rounded features (sc2bench/models/segmentation/deeplabv3.py:44-104): 2e-2 rel. L2."""
    from sc2bench_amd import dense
    torch.manual_seed(N + H)
    for head, cin in ((dense.DeepLabHead(2048, 21), 2048), (dense.FCNHead(1024, 21), 1024)):
        head.eval()
        _randomise_norms(head)
        x = (torch.randn(N, cin, H, W) * 0.5).to(torch.bfloat16)
        with torch.no_grad():
            ref = head.float()(x.float())
        head = head.to(dev).to(torch.bfloat16)
        assert dense.HipDenseHead.supported(head)
        hd = dense.HipDenseHead(head)
        with torch.no_grad():
            got = hd(x.to(dev).contiguous(memory_format=torch.channels_last))
        assert got.shape == ref.shape
        _close(got, ref, 2e-2)


def test_conv2d_fwd_dilation_vs_torch(S, dev):
    """sc2_conv2d_fwd with the descriptor's dilation (the generic tile's Cfg::DIL instantiation): dilation 2 / 4 with padding ==
    dilation (torchvision's dilated layer3 / layer4), 12 on a map smaller than the reach of its taps, unequal padding, stride 2;
    tap-major and slab-major K; against F.conv2d on the bf16-rounded operands."""
    import torch.nn.functional as F
    hip = S.hip
    torch.manual_seed(1)
    for (cin, cout, k, stride, pad, dil, H, W, order) in [(64, 128, 3, 1, 2, 2, 19, 23, hip.K_TAP_MAJOR), (64, 256, 3, 1, 4, 4, 17, 9, hip.K_SLAB_MAJOR),
                                                          (96, 128, 3, 1, 12, 12, 15, 20, hip.K_SLAB_MAJOR), (32, 128, 3, 2, 1, 3, 21, 21, hip.K_TAP_MAJOR),
                                                          (40, 128, 2, 1, 0, 5, 14, 14, hip.K_TAP_MAJOR)]:
        x = torch.randn(2, cin, H, W).to(torch.bfloat16)
        w = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).to(torch.bfloat16)
        ref = F.conv2d(x.float(), w.float(), stride=stride, padding=pad, dilation=dil)
        got = hip.conv2d_fwd(x.to(dev).permute(0, 2, 3, 1).contiguous(), hip.pack_conv_weight(w.to(dev), order), cout, k, k, stride, pad,
                             k_order=order, dilation=dil)
        assert tuple(got.shape) == (2, ref.shape[2], ref.shape[3], cout)
        _close(got.permute(0, 3, 1, 2), ref, 6e-3)


def test_fpn_on_hip_kernels_vs_torch_f32(S, dev):
    """The feature pyramid's lateral 1x1 and output 3x3 convs (with bias) on the library's kernels vs the torch module in f32."""
    from collections import OrderedDict
    from sc2bench_amd import dense
    torch.manual_seed(3)
    fpn = dense.FeaturePyramidNetwork([256, 512, 1024, 2048], 256, extra_blocks=dense.LastLevelMaxPool()).eval()
    _randomise_norms(fpn)
    shapes = [(256, 40, 52), (512, 20, 26), (1024, 10, 13), (2048, 5, 7)]
    feats = OrderedDict((str(i), (torch.randn(2, c, h, w) * 0.5).to(torch.bfloat16)) for i, (c, h, w) in enumerate(shapes))
    with torch.no_grad():
        ref = fpn.float()(OrderedDict((k, v.float()) for k, v in feats.items()))
    fpn = fpn.to(dev).to(torch.bfloat16)
    hd = dense.HipDenseHead(fpn)
    with torch.no_grad():
        results, names = hd.fpn(OrderedDict((k, v.to(dev).contiguous(memory_format=torch.channels_last)) for k, v in feats.items()))
    assert names == list(feats.keys())
    for name, r in zip(names, results):
        _close(r, ref[name], 2e-2)
