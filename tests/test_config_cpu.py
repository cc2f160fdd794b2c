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
def test_entropic_student_config_builds_hip_model_and_criterion(S):
    from sc2bench_amd import config as C, training as T
    cfg = C.load_yaml_file(ES)
    C.import_dependencies(cfg['dependencies'])
    student = C.build_model(cfg['models']['student_model'])
    teacher = C.build_model(cfg['models']['teacher_model'])
    assert isinstance(student, S.SplittableResNet) and isinstance(student.bottleneck_layer, S.FPBasedResNetBottleneck)
    assert student.analyzes_after_compress and isinstance(student.analyzers[0], S.FileSizeAnalyzer)
    assert cfg['models']['student_model']['dst_ckpt'].endswith('ilsvrc2012-splittable_resnet50-fp-beta0.08_from_resnet50.pt')
    ds = cfg['datasets']['ilsvrc2012/train']
    assert isinstance(ds, C.SyntheticImageFolder) and ds[0][0].shape == (3, 224, 224)
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
