"""The reference's YAML files load unchanged through sc2bench_amd.config (CPU; skipped where /root/reference is absent)."""
import glob
import os

import pytest
import torch

REF = '/root/reference/configs'
ES = os.path.join(REF, 'ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_resnet50.yaml')
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present (GPU box)')


@needs_ref
def test_every_reference_config_parses(S):
    from sc2bench_amd import config as C
    files = sorted(glob.glob(os.path.join(REF, '**', '*.yaml'), recursive=True))
    assert len(files) >= 180
    for f in files:
        cfg = C.load_yaml_file(f)
        assert 'models' in cfg and 'datasets' in cfg


@needs_ref
def test_entropic_student_config_builds_hip_model_and_criterion(S, monkeypatch, recwarn):
    from sc2bench_amd import config as C, training as T
    cfg = C.load_yaml_file(ES)
    C.import_dependencies(cfg['dependencies'])
    student = C.build_model(cfg['models']['student_model'])
    teacher = C.build_model(cfg['models']['teacher_model'])
    assert isinstance(student, S.SplittableResNet) and isinstance(student.bottleneck_layer, S.FPBasedResNetBottleneck)
    assert student.analyzes_after_compress and isinstance(student.analyzers[0], S.FileSizeAnalyzer)
    assert cfg['models']['student_model']['dst_ckpt'].endswith('ilsvrc2012-splittable_resnet50-fp-beta0.08_from_resnet50.pt')
    ds = cfg['datasets']['ilsvrc2012/train']
    assert isinstance(ds, C.ImageFolder)
    with pytest.raises(FileNotFoundError):      # ~/datasets/ilsvrc2012 is absent: never silently random data
        len(ds)
    monkeypatch.setenv('SC2_SYNTHETIC_DATA', '1')
    assert ds[0][0].shape == (3, 224, 224) and len(ds) > 0
    stage1 = cfg['train']['stage1']
    crit = T.build_criterion(stage1['criterion'])
    assert sorted(crit.terms.keys()) == ['bpp', 'layer1', 'layer2', 'layer3', 'layer4'] and crit.weights['bpp'] == 0.08
    assert isinstance(crit.terms['bpp'], S.BppLoss) and crit.terms['bpp'].reduction == 'sum'
    seq = T.redesign_model(student, stage1['student']['sequential'])
    assert [n for n, _ in seq.named_children()] == ['bottleneck_layer', 'layer2', 'layer3', 'layer4']
    hooks = T.ForwardHookManager(seq, stage1['student']['forward_hook'])
    assert len(hooks.handles) == 5
    T.freeze(student, stage1['student']['frozen_modules'])
    trainable = sum(p.numel() for p in student.parameters() if p.requires_grad)
    assert trainable == 1304168 + sum(p.numel() for p in student.fc.parameters())
    C.overwrite_config(cfg, {'train': {'stage1': {'num_epochs': 1}}})
    assert cfg['train']['stage1']['num_epochs'] == 1 and 'optimizer' in cfg['train']['stage1']
    assert teacher.layer1 is not None
    # the teacher block asks for ImageNet weights: without a local file the builder says so
    assert any('RANDOMLY INITIALISED' in str(w.message) for w in recwarn.list)


def test_image_folder_and_ckpt_loading(tmp_path, monkeypatch):
    from PIL import Image
    from sc2bench_amd import config as C, ckpt
    from sc2bench_amd.resnet import resnet50
    for c, colour in (('cat', (255, 0, 0)), ('dog', (0, 255, 0))):
        os.makedirs(tmp_path / 'val' / c)
        for i in range(2):
            Image.new('RGB', (8, 6), colour).save(tmp_path / 'val' / c / '{}.png'.format(i))
    ds = C.ImageFolder(str(tmp_path / 'val'))
    assert len(ds) == 4 and ds.classes == ['cat', 'dog'] and ds[3][1] == 1 and ds[0][0].size == (8, 6)
    with pytest.raises(RuntimeError):
        ckpt.load_ckpt('https://example.org/model.pt')
    assert ckpt.load_ckpt(str(tmp_path / 'missing.pt')) == (None, None)
    # weights= resolves to $SC2_PRETRAINED_DIR/<name>.pth; SC2_STRICT_WEIGHTS turns the warning into an error
    m = resnet50()
    torch.save(m.state_dict(), tmp_path / 'resnet50.pth')
    monkeypatch.setenv('SC2_PRETRAINED_DIR', str(tmp_path))
    m2 = resnet50(weights='ResNet50_Weights.IMAGENET1K_V1')
    assert torch.equal(m2.fc.weight, m.fc.weight)
    monkeypatch.setenv('SC2_PRETRAINED_DIR', str(tmp_path / 'nowhere'))
    monkeypatch.setenv('SC2_STRICT_WEIGHTS', '1')
    with pytest.raises(FileNotFoundError):
        resnet50(weights='ResNet50_Weights.IMAGENET1K_V1')
    # a checkpoint that needs arbitrary unpickling is refused by default
    class Evil(object):
        def __reduce__(self):
            return (print, ('pwned',))
    torch.save({'model': m.state_dict(), 'args': Evil()}, tmp_path / 'evil.pt')
    with pytest.raises(RuntimeError):
        ckpt.load_ckpt(str(tmp_path / 'evil.pt'), model=m)
    import argparse
    ckpt.save_ckpt(m, None, None, 0.5, argparse.Namespace(a=1), str(tmp_path / 'ok.pt'))
    best, args = ckpt.load_ckpt(str(tmp_path / 'ok.pt'), model=m)
    assert best == 0.5 and args.a == 1


def test_criterion_arithmetic():
    from sc2bench_amd import training as T
    cfg = {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': {
        'feat': {'criterion': {'key': 'MSELoss', 'kwargs': {'reduction': 'sum'}},
                 'criterion_wrapper': {'key': 'SimpleLossWrapper', 'kwargs': {
                     'input': {'is_from_teacher': False, 'module_path': 'a', 'io': 'output'},
                     'target': {'is_from_teacher': True, 'module_path': 'b', 'io': 'output'}}}, 'weight': 2.0},
        'bpp': {'criterion': {'key': 'BppLoss', 'kwargs': {'entropy_module_path': 'eb', 'reduction': 'sum'}}, 'weight': 0.5}}}}
    crit = T.build_criterion(cfg)
    x, y = torch.ones(2, 3, 2, 2), torch.zeros(2, 3, 2, 2)
    lik = torch.full((2, 3, 2, 2), 0.25)
    s_io = {'a': {'output': x}, 'eb': {'output': (x, lik)}}
    t_io = {'b': {'output': y}}
    assert abs(crit(s_io, t_io).item() - (2.0 * 24 + 0.5 * 2 * 24)) < 1e-5
