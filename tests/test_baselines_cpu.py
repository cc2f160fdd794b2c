"""CPU: the reference's codec-feature-compression baseline (BASELINE config 1:
configs/ilsvrc2012/feature_compression/jpeg-resnet50.yaml -test_only on CPU, 100 images) through this build's config
loader, wrappers and transforms, against the oracle's restatement of the same arithmetic; plus the host transforms the
input-compression config uses (AdaptivePad, Resize / CenterCrop / ToTensor / Normalize)."""
import os

import numpy as np
import pytest
import torch

REF = '/root/reference/configs'
JPEG_CFG = os.path.join(REF, 'ilsvrc2012/feature_compression/jpeg-resnet50.yaml')
FP_CFG = os.path.join(REF, 'ilsvrc2012/input_compression/factorized_prior-resnet50.yaml')
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present (GPU box)')


def _image_folder(root, n, size=(80, 64)):
    from PIL import Image
    rng = np.random.RandomState(0)
    for i in range(n):
        d = os.path.join(root, 'val', 'class{:02d}'.format(i % 10))
        os.makedirs(d, exist_ok=True)
        base = rng.randint(0, 255, size=(size[1] // 8, size[0] // 8, 3)).astype(np.uint8)      # blocky, compressible
        img = Image.fromarray(base, 'RGB').resize(size, 0)
        img.save(os.path.join(d, 'img{:03d}.png'.format(i)))


def test_pil_tensor_module_matches_oracle():
    from oracle import cpu_ref_input as RI
    from sc2bench_amd.transforms import PILTensorModule
    torch.manual_seed(0)
    for C, H, W in ((512, 28, 28), (8, 17, 23), (5, 9, 9), (3, 16, 16), (1, 8, 8)):
        x = torch.randn(C, H, W).abs() * 3 + 0.1
        m = PILTensorModule(returns_file_size=True, format='JPEG', quality=90)
        y, size = m(x)
        y_ref, size_ref = RI.pil_tensor_module(x, format='JPEG', quality=90)
        assert y.shape == x.shape and torch.equal(y, y_ref) and size == size_ref
        assert isinstance(size, (int, float)) and size > 0
    # a lossless codec keeps what the 8-bit normalisation keeps: x ~ (round(255 (x - min) / max) / 255) max + min
    x = torch.rand(6, 12, 12) + 0.5
    y = PILTensorModule(format='PNG')(x)
    for g in range(2):
        xs = x[3 * g:3 * g + 3]
        mn, mx = xs.min(), xs.max()
        q = ((xs - mn) / mx).mul(255).byte().float() / 255 * mx + mn
        assert torch.allclose(y[3 * g:3 * g + 3], q, atol=1e-6)


def test_adaptive_pad_and_host_transforms():
    from oracle import cpu_ref_input as RI
    from sc2bench_amd import transforms as T
    x = torch.rand(3, 224, 224)
    for kw in (dict(fill=0, factor=64), dict(fill=0, factor=128), dict(fill=1, factor=64, padding_position='equal_side')):
        y = T.AdaptivePad(**kw)(x)
        assert torch.equal(y, RI.adaptive_pad(x, **kw))
    y = T.AdaptivePad(fill=0, factor=64)(x)
    assert y.shape == (3, 256, 256) and torch.equal(y[:, :224, :224], x) and float(y[:, 224:].abs().max()) == 0
    assert T.AdaptivePad(factor=64)(torch.rand(3, 128, 192)).shape == (3, 128, 192)
    y, hw = T.AdaptivePad(factor=32, returns_org_patch_size=True)(torch.rand(2, 3, 33, 65))
    assert y.shape == (2, 3, 64, 96) and hw == (33, 65)
    from PIL import Image
    img = Image.fromarray(np.random.RandomState(1).randint(0, 255, size=(300, 500, 3)).astype(np.uint8), 'RGB')
    r = T.Resize(256)(img)
    assert r.size == (426, 256)            # (w, h): shorter side 256, int(256 * 500 / 300) = 426
    c = T.CenterCrop([224, 224])(r)
    assert c.size == (224, 224)
    t = T.ToTensor()(c)
    assert t.shape == (3, 224, 224) and t.dtype == torch.float32 and 0 <= float(t.min()) and float(t.max()) <= 1
    assert torch.equal(t, torch.from_numpy(np.array(c)).permute(2, 0, 1).float() / 255)
    n = T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])(t)
    assert torch.allclose(n[1], (t[1] - 0.456) / 0.224)
    assert T.CenterCrop(224)(torch.rand(3, 256, 256)).shape == (3, 224, 224)


@needs_ref
def test_config1_jpeg_feature_compression_test_only_cpu(tmp_path, monkeypatch):
    """100 images, batch size 1, CPU, through sc2bench_amd.evaluation.test_only; the wrapper's logits and file sizes equal
    the oracle's restatement of CodecFeatureCompressionClassifier.forward on the same model and samples."""
    from oracle import cpu_ref_input as RI
    from sc2bench_amd import config as C, evaluation as E
    import sc2bench_amd as S
    _image_folder(str(tmp_path), 100)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = C.load_yaml_file(JPEG_CFG)
    cfg['datasets']['ilsvrc2012/val'].root = str(tmp_path / 'val')
    res = E.test_only(cfg, torch.device('cpu'), max_samples=100, num_workers=0)
    assert res['samples'] == 100 and 0.0 <= res['acc1'] <= res['acc5'] <= 100.0
    assert len(res['analysis']) == 1 and res['analysis'][0]['count'] == 100 and res['analysis'][0]['unit'] == 'KB'
    assert res['analysis'][0]['mean'] > 1.0
    # same model, three samples, against the oracle's forward
    torch.manual_seed(0)
    model = C.build_model(cfg['models']['model'], torch.device('cpu')).eval()
    assert isinstance(model, S.CodecFeatureCompressionClassifier) and isinstance(model.analyzers[0], S.FileSizeAccumulator)
    assert [n for n, _ in model.encoder.named_children()] == ['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2']
    assert [n for n, _ in model.decoder.named_children()] == ['layer3', 'layer4', 'avgpool']
    ds = cfg['datasets']['ilsvrc2012/val']
    model.activate_analysis()
    with torch.inference_mode():
        for i in (0, 41, 99):
            x, _ = ds[i]
            out = model(x.unsqueeze(0))
            ref_out, sizes = RI.codec_feature_compression_forward(
                model.encoder, lambda t: RI.pil_tensor_module(t, format='JPEG', quality=90), model.decoder,
                model.classifier, x.unsqueeze(0))
            assert torch.equal(out, ref_out)
            assert model.analyzers[0].file_size_list[-1] == sizes[0] / 1024
    assert out.shape == (1, 1000)


@needs_ref
def test_config3_parses_into_real_transforms_and_wrapper(monkeypatch):
    """The factorized-prior input-compression config builds real transforms; the model itself needs a HIP device
    (tests/test_gpu_input_compression.py)."""
    from sc2bench_amd import config as C, transforms as T
    cfg = C.load_yaml_file(FP_CFG)
    tr = cfg['datasets']['ilsvrc2012/val'].transform
    assert [type(t).__name__ for t in tr.transforms] == ['Resize', 'CenterCrop', 'ToTensor', 'AdaptivePad']
    assert tr.transforms[3].factor == 64 and tr.transforms[3].fill == 0
    post = cfg['models']['model']['kwargs']['post_transform']
    assert [type(t).__name__ for t in post.transforms] == ['CenterCrop', 'Normalize']
    x = torch.rand(3, 256, 256)
    assert post(x).shape == (3, 224, 224)
    assert cfg['models']['model']['compression_model'] == {'key': 'bmshj2018_factorized',
                                                           'kwargs': {'pretrained': True, 'quality': 8, 'metric': 'mse'}}
    assert cfg['models']['model']['kwargs']['analysis_config']['analyzes_after_compress'] is True
