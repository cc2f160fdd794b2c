"""Loads the reference's experiment YAML files unchanged.

The reference's `configs/**.yaml` use four torchdistill YAML tags: `!join`, `!import_get`,
`!import_call`, `!getattr`, evaluated eagerly at load time (script/task/image_classification.py:207), and dotted keys
into packages that are not installed here (`torchvision`, `torchdistill`) or that this package replaces
(`sc2bench`).  This loader resolves

* `sc2bench.*`      -> the registries of this package (same keys: layers, backbones, losses, analyzers),
* `torchvision.datasets.*` -> this module's `ImageFolder` (root/<class>/<image>; a missing directory raises on first
  use unless SC2_SYNTHETIC_DATA=1 opts in to random images),
* `torchvision.models.resnet.*Weights` -> an inert enum (pretrained weights need the network),
* `torchvision.transforms.{Compose,Resize,CenterCrop,ToTensor,Normalize,RandomResizedCrop,RandomHorizontalFlip}` ->
  `sc2bench_amd.transforms` (PIL + torch),
* other `torchvision.*`, `torchdistill.*`, anything else that cannot be imported -> a recording placeholder,

so every config parses, and `models.student_model` builds the HIP-backed model from the very same block.
"""
import importlib
import os

import torch
import yaml


class Placeholder(object):
    """Stands in for an object of a package that is not available offline; records how it was built."""

    def __init__(self, key, args=(), kwargs=None):
        self.key = key
        self.args = tuple(args)
        self.kwargs = dict(kwargs or {})

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return Placeholder('{}.{}'.format(self.key, name))

    def __call__(self, *args, **kwargs):
        return Placeholder(self.key, args, kwargs)

    def __repr__(self):
        return 'Placeholder({})'.format(self.key)


class SyntheticImageFolder(torch.utils.data.Dataset):
    """Seeded random images with random labels, ImageNet-normalised: plumbing / throughput runs only."""

    def __init__(self, root=None, transform=None, num_samples=1024, num_classes=1000, image_size=224, **kwargs):
        self.root = root
        self.transform = transform
        self.num_samples = num_samples
        self.num_classes = num_classes
        self.image_size = image_size
        self.mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        self.std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def __len__(self):
        return self.num_samples

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(index)
        x = torch.rand(3, self.image_size, self.image_size, generator=g)
        return (x - self.mean) / self.std, int(torch.randint(0, self.num_classes, (1,), generator=g))


IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


class ImageFolder(torch.utils.data.Dataset):
    """`torchvision.datasets.ImageFolder(root, transform)`: root/<class>/<image>, classes sorted by name, label = class
    index, images opened with PIL and converted to RGB.  The directory is scanned on first use, so a config whose
    dataset directory does not exist still PARSES; using such a dataset raises, unless SC2_SYNTHETIC_DATA=1 opts in to
    `SyntheticImageFolder` (random pixels, random labels -- never silently)."""

    def __init__(self, root=None, transform=None, target_transform=None, **kwargs):
        self.root = None if root is None else os.path.expanduser(str(root))
        self.transform = transform
        self.target_transform = target_transform
        self.kwargs = kwargs
        self._samples = None
        self._synthetic = None

    def _scan(self):
        if self._samples is not None or self._synthetic is not None:
            return
        if self.root is not None and os.path.isdir(self.root):
            classes = sorted(d for d in os.listdir(self.root) if os.path.isdir(os.path.join(self.root, d)))
            samples = []
            for ci, c in enumerate(classes):
                for dirpath, _, files in sorted(os.walk(os.path.join(self.root, c), followlinks=True)):
                    for f in sorted(files):
                        if f.lower().endswith(IMG_EXTENSIONS):
                            samples.append((os.path.join(dirpath, f), ci))
            if not samples:
                raise FileNotFoundError('ImageFolder: no images under {}'.format(self.root))
            self.classes, self._samples = classes, samples
            return
        if os.environ.get('SC2_SYNTHETIC_DATA') == '1':
            import logging
            logging.getLogger(__name__).warning('dataset directory {} not found: SC2_SYNTHETIC_DATA=1 -> random pixels, '
                                                'random labels'.format(self.root))
            self._synthetic = SyntheticImageFolder(self.root, **{k: v for k, v in self.kwargs.items()
                                                                 if k in ('num_samples', 'num_classes', 'image_size')})
            return
        raise FileNotFoundError('ImageFolder: dataset directory {} does not exist (set SC2_SYNTHETIC_DATA=1 to run on '
                                'random images with random labels instead)'.format(self.root))

    def __len__(self):
        self._scan()
        return len(self._synthetic) if self._synthetic is not None else len(self._samples)

    def __getitem__(self, index):
        self._scan()
        if self._synthetic is not None:
            return self._synthetic[index]
        from PIL import Image
        path, target = self._samples[index]
        with open(path, 'rb') as f:
            img = Image.open(f).convert('RGB')
        if self.transform is not None and not isinstance(self.transform, Placeholder):
            img = self.transform(img)
        if self.target_transform is not None and not isinstance(self.target_transform, Placeholder):
            target = self.target_transform(target)
        return img, target


class _WeightsEnum(object):
    """Inert stand-in for torchvision's `ResNet50_Weights` etc. (`!getattr [*, 'IMAGENET1K_V1']` must resolve)."""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return '{}.{}'.format(self._name, item)


def _sc2bench_attr(key):
    import sc2bench_amd as S
    from . import loss as loss_mod
    name = key.split('.')[-1]
    for registry in (S.LAYER_CLASS_DICT, S.LAYER_FUNC_DICT, S.BACKBONE_CLASS_DICT, S.BACKBONE_FUNC_DICT,
                     S.ANALYZER_CLASS_DICT, loss_mod.MIDDLE_LEVEL_LOSS_DICT):
        if name in registry:
            return registry[name]
    if hasattr(S, name):
        return getattr(S, name)
    from . import transforms as tr, wrapper as wr, compression as cm, dense as dn
    for registry in (tr.CODEC_TRANSFORM_MODULE_DICT, tr.MISC_TRANSFORM_MODULE_DICT, wr.WRAPPER_CLASS_DICT,
                     cm.COMPRESSION_MODEL_CLASS_DICT, cm.COMPRESSION_MODEL_FUNC_DICT, dn.DETECTION_MODEL_FUNC_DICT,
                     dn.SEGMENTATION_MODEL_FUNC_DICT):
        if name in registry:
            return registry[name]
    return Placeholder(key)


def resolve(key):
    """Dotted name -> object, with the substitutions listed in the module docstring."""
    if key.startswith('sc2bench.'):
        return _sc2bench_attr(key)
    if key.startswith('torchvision.datasets.'):
        return ImageFolder
    if key.startswith('torchvision.models.') and key.endswith('_Weights'):
        return _WeightsEnum(key)
    if key.startswith('torchvision.transforms.'):
        from . import transforms as tr
        name = key.split('.')[-1]
        if name in tr.TORCHVISION_TRANSFORM_DICT:
            return tr.TORCHVISION_TRANSFORM_DICT[name]
    module_name, _, attr = key.rpartition('.')
    try:
        return getattr(importlib.import_module(module_name), attr)
    except Exception:
        return Placeholder(key)


class _Loader(yaml.SafeLoader):
    pass


def _join(loader, node):
    return ''.join(str(v) for v in loader.construct_sequence(node, deep=True))


def _import_get(loader, node):
    entry = loader.construct_mapping(node, deep=True)
    return resolve(entry['key'])


def _import_call(loader, node):
    entry = loader.construct_mapping(node, deep=True)
    init = entry.get('init') or dict()
    args = init.get('args') or list()
    kwargs = init.get('kwargs') or dict()
    target = resolve(entry['key'])
    return target(*args, **kwargs)


def _getattr(loader, node):
    obj, name = loader.construct_sequence(node, deep=True)
    return getattr(obj, name)


_Loader.add_constructor('!join', _join)
_Loader.add_constructor('!import_get', _import_get)
_Loader.add_constructor('!import_call', _import_call)
_Loader.add_constructor('!getattr', _getattr)


def load_yaml_file(path):
    """torchdistill.common.yaml_util.load_yaml_file: the config dict with every tag evaluated."""
    with open(os.path.expanduser(path)) as f:
        return yaml.load(f, Loader=_Loader)


def overwrite_config(org_config, sub_config):
    """`--json` deep overwrite (sc2bench/common/config_util.py:1-17)."""
    for key, value in sub_config.items():
        if key in org_config and isinstance(value, dict) and isinstance(org_config[key], dict):
            overwrite_config(org_config[key], value)
        else:
            org_config[key] = value


def import_dependencies(dependencies):
    """`dependencies: [{name: sc2bench.models}, ...]` -- registration by import; sc2bench names map to this package."""
    for dep in dependencies or list():
        name = dep['name'] if isinstance(dep, dict) else dep
        if name.startswith('sc2bench'):
            importlib.import_module('sc2bench_amd')
        else:
            try:
                importlib.import_module(name)
            except Exception:
                pass


def build_model(model_config, device='cpu'):
    """models.{teacher_model, student_model, model} block -> nn.Module (torchvision / sc2bench registries); a block with a
    `classification_model` entry is a wrapped baseline (sc2bench/models/wrapper.py:343-370)."""
    import sc2bench_amd as S
    from .resnet import RESNET_FUNC_DICT
    if 'classification_model' in model_config:
        from .wrapper import get_wrapped_classification_model
        return get_wrapped_classification_model(model_config, device)
    key = model_config['key']
    kwargs = dict(model_config.get('kwargs') or {})
    from . import dense as dn
    for registry in (dn.DETECTION_MODEL_FUNC_DICT, dn.SEGMENTATION_MODEL_FUNC_DICT):
        if key in registry:
            return registry[key](**kwargs)
    if key in S.MODEL_DICT:
        if kwargs.get('weights') is not None:   # splittable_resnet & co. forward **kwargs to the torchvision builder
            import logging
            logging.getLogger(__name__).warning('model `{}`: weights={!r} is passed on to the ResNet builder, which loads '
                                                '$SC2_PRETRAINED_DIR/<name>.pth or warns'.format(key, kwargs['weights']))
        return S.MODEL_DICT[key](**kwargs)
    if key in RESNET_FUNC_DICT:
        return RESNET_FUNC_DICT[key](**kwargs)
    raise KeyError('model key `{}` is not available in this build (have {})'.format(
        key, sorted(list(S.MODEL_DICT) + list(RESNET_FUNC_DICT))))
