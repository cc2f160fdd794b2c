"""Transforms the reference's configs name: the sc2bench ones (`sc2bench.transforms.codec`, `sc2bench.transforms.misc`)
and the handful of torchvision ones the BASELINE configs build with `!import_call` (torchvision is not installed in this
image; `config.resolve` maps `torchvision.transforms.<Name>` here).

* `PILTensorModule` -- sc2bench/transforms/codec.py:114-186: a feature tensor [C,H,W] goes through an image codec three
  channels at a time (config 1: JPEG quality 90, configs/ilsvrc2012/feature_compression/jpeg-resnet50.yaml:42-48).  The
  normalisation is the reference's `(x - min) / max` and `* max + min` (codec.py:159,170) -- not a min-max scaling, kept
  as is because the reported sizes and accuracies depend on it.
* `PILImageModule` -- codec.py:79-111.
* `AdaptivePad` -- sc2bench/transforms/misc.py:105-154 (tests 'equal_side', otherwise pads right and bottom).
* `Compose`, `Resize`, `CenterCrop`, `ToTensor`, `Normalize`, `RandomResizedCrop`, `RandomHorizontalFlip` -- torchvision
  semantics on PIL images / tensors, PIL + torch only.
These run on the host (PIL codecs are host code in the reference too); they are callers of the hot path, not part of it.
"""
import math
import random
from io import BytesIO

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .analysis import get_binary_object_size

CODEC_TRANSFORM_MODULE_DICT = dict()
MISC_TRANSFORM_MODULE_DICT = dict()
TORCHVISION_TRANSFORM_DICT = dict()


def register_codec_transform_module(cls):
    CODEC_TRANSFORM_MODULE_DICT[cls.__name__] = cls
    return cls


def register_misc_transform_module(cls):
    MISC_TRANSFORM_MODULE_DICT[cls.__name__] = cls
    return cls


def _tv(cls):
    TORCHVISION_TRANSFORM_DICT[cls.__name__] = cls
    return cls


# --------------------------------------------------------------------------------------------------------- functional
def _is_pil(x):
    from PIL import Image
    return isinstance(x, Image.Image)


def to_pil_image(pic):
    """float CHW in [0, 1] (-> mul(255).byte(), as torchvision: out-of-range values wrap) or uint8 CHW -> PIL L / RGB."""
    from PIL import Image
    if pic.dim() == 2:
        pic = pic.unsqueeze(0)
    if pic.is_floating_point():
        pic = pic.mul(255).byte()
    arr = np.transpose(pic.cpu().numpy(), (1, 2, 0))
    if arr.shape[2] == 1:
        return Image.fromarray(np.ascontiguousarray(arr[:, :, 0]), mode='L')
    if arr.shape[2] == 3:
        return Image.fromarray(np.ascontiguousarray(arr), mode='RGB')
    raise ValueError('to_pil_image: {} channels'.format(arr.shape[2]))


def to_tensor(pic):
    """8-bit PIL image (or HWC uint8 ndarray) -> float CHW tensor / 255."""
    arr = np.array(pic, copy=True)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


def pad(x, padding, fill=0, padding_mode='constant'):
    """torchvision.transforms.functional.pad for tensors [..., H, W] and PIL images; padding = int | [lr, tb] | [l, t, r, b]."""
    if isinstance(padding, int):
        padding = [padding] * 4
    elif len(padding) == 2:
        padding = [padding[0], padding[1], padding[0], padding[1]]
    left, top, right, bottom = padding
    if _is_pil(x):
        from PIL import ImageOps
        assert padding_mode == 'constant'
        return ImageOps.expand(x, border=(left, top, right, bottom), fill=fill)
    if padding_mode == 'constant':
        return F.pad(x, [left, right, top, bottom], mode='constant', value=fill)
    return F.pad(x, [left, right, top, bottom], mode=padding_mode)


def _size_hw(x):
    if _is_pil(x):
        return x.size[1], x.size[0]
    return x.shape[-2], x.shape[-1]


# --------------------------------------------------------------------------------------------------------- torchvision
@_tv
class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x

    def __repr__(self):
        return 'Compose({})'.format(self.transforms)


INTERPOLATION = {'nearest': 0, 'lanczos': 1, 'bilinear': 2, 'bicubic': 3, 'box': 4, 'hamming': 5}


@_tv
class Resize(nn.Module):
    """int size: the shorter side becomes `size` (aspect kept); (h, w): exact.  PIL images (bilinear by default)."""

    def __init__(self, size, interpolation='bilinear', **kwargs):
        super().__init__()
        self.size = size
        self.interpolation = INTERPOLATION.get(interpolation, 2) if isinstance(interpolation, str) else 2

    def forward(self, img):
        h, w = _size_hw(img)
        if isinstance(self.size, int) or len(self.size) == 1:
            s = self.size if isinstance(self.size, int) else self.size[0]
            short, long_ = (w, h) if w <= h else (h, w)
            new_short, new_long = s, int(s * long_ / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = self.size
        if _is_pil(img):
            return img.resize((nw, nh), self.interpolation)
        x = img.unsqueeze(0) if img.dim() == 3 else img
        x = F.interpolate(x.float(), size=(nh, nw), mode='bilinear', align_corners=False, antialias=True)
        return x.squeeze(0) if img.dim() == 3 else x


@_tv
class CenterCrop(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def forward(self, img):
        th, tw = self.size
        h, w = _size_hw(img)
        if tw > w or th > h:     # torchvision pads with zeros first
            pl, pt = max((tw - w) // 2, 0), max((th - h) // 2, 0)
            img = pad(img, [pl, pt, max((tw - w + 1) // 2, 0), max((th - h + 1) // 2, 0)])
            h, w = _size_hw(img)
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        if _is_pil(img):
            return img.crop((left, top, left + tw, top + th))
        return img[..., top:top + th, left:left + tw]


@_tv
class ToTensor(object):
    def __call__(self, pic):
        return to_tensor(pic)


@_tv
class Normalize(nn.Module):
    def __init__(self, mean, std, inplace=False):
        super().__init__()
        self.mean, self.std = list(mean), list(std)

    def forward(self, x):
        # (the two constant vectors live on the input's device once: building them per call is a blocking host-to-device copy,
        #  4 ms each with a busy device queue -- it was 8 of the 12 ms of a pipelined fp_input step, tools/host_prof_workload.py)
        key = (x.dtype, x.device)
        cached = self.__dict__.get('_consts')
        if cached is None or cached[0] != key or cached[3] != (tuple(self.mean), tuple(self.std)):
            # built OUTSIDE inference mode whatever the caller's mode: evaluate() runs under torch.inference_mode, and an inference
            # tensor cached there cannot be saved for backward by a later grad-mode call (codec training through
            # NeuralInputCompressionClassifier's post_transform)
            with torch.inference_mode(False):
                cached = (key, torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1),
                          torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1), (tuple(self.mean), tuple(self.std)))
            self.__dict__['_consts'] = cached
        return (x - cached[1]) / cached[2]

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_consts', None)          # device tensors of a cache do not belong in a pickle
        return state


@_tv
class RandomHorizontalFlip(nn.Module):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, img):
        if random.random() >= self.p:
            return img
        if _is_pil(img):
            from PIL import Image
            return img.transpose(Image.FLIP_LEFT_RIGHT)
        return img.flip(-1)


@_tv
class RandomResizedCrop(nn.Module):
    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation='bilinear', **kwargs):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio = scale, ratio
        self.resize = Resize(self.size, interpolation if interpolation is not None else 'bilinear')

    def forward(self, img):
        h, w = _size_hw(img)
        area = h * w
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        top = left = 0
        ch, cw = h, w
        for _ in range(10):
            target = area * random.uniform(self.scale[0], self.scale[1])
            aspect = math.exp(random.uniform(*log_ratio))
            tw, th = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if 0 < tw <= w and 0 < th <= h:
                top, left, ch, cw = random.randint(0, h - th), random.randint(0, w - tw), th, tw
                break
        else:   # central crop at the closest valid ratio
            in_ratio = w / h
            if in_ratio < min(self.ratio):
                cw, ch = w, int(round(w / min(self.ratio)))
            elif in_ratio > max(self.ratio):
                ch, cw = h, int(round(h * max(self.ratio)))
            top, left = (h - ch) // 2, (w - cw) // 2
        img = img.crop((left, top, left + cw, top + ch)) if _is_pil(img) else img[..., top:top + ch, left:left + cw]
        return self.resize(img)


# --------------------------------------------------------------------------------------------------------- sc2bench.misc
@register_misc_transform_module
class AdaptivePad(nn.Module):
    """Pads an image tensor so that both sides become multiples of `factor` (misc.py:105-154)."""

    def __init__(self, fill=0, padding_position='hw', padding_mode='constant', factor=128, returns_org_patch_size=False):
        super().__init__()
        self.fill = fill
        self.padding_position = padding_position
        self.padding_mode = padding_mode
        self.factor = factor
        self.returns_org_patch_size = returns_org_patch_size

    def forward(self, x):
        # written against the behaviour of misc.py:141-154 (not its text): each side grows by what it lacks to the next
        # multiple of `factor`; 'equal_side' hands torchvision's two-value form HALF of that per side (so an odd remainder
        # ends one short of a multiple -- the reference's behaviour, kept), anything else pads right and bottom only
        org_h, org_w = (int(v) for v in x.shape[-2:])
        lack_h, lack_w = -org_h % self.factor, -org_w % self.factor
        if self.padding_position == 'equal_side':
            borders = [lack_w // 2, lack_h // 2]                # left = right, top = bottom
        else:
            borders = [0, 0, lack_w, lack_h]                    # left, top, right, bottom
        x = pad(x, borders, self.fill, self.padding_mode)
        return (x, (org_h, org_w)) if self.returns_org_patch_size else x


@register_misc_transform_module
class ClearTargetTransform(nn.Module):
    def forward(self, sample, *args):
        return sample, list()


def default_collate_w_pil(batch):
    """torchdistill's `default_collate_w_pil`: like default_collate, but PIL images are kept as a list."""
    from torch.utils.data import default_collate
    elem = batch[0]
    if _is_pil(elem):
        return list(batch)
    if isinstance(elem, (tuple, list)):
        return [default_collate_w_pil(list(samples)) for samples in zip(*batch)]
    return default_collate(batch)


# --------------------------------------------------------------------------------------------------------- sc2bench.codec
def _pil_codec_round_trip(pil_img, save_kwargs, open_kwargs):
    """image -> codec bytes in memory -> reopened image: (image, byte count of the coded file)."""
    from PIL import Image
    coded = BytesIO()
    pil_img.save(coded, **save_kwargs)
    n_bytes = coded.tell()          # before reopening: the decoder moves the position
    return Image.open(coded, **open_kwargs), n_bytes


@register_codec_transform_module
class PILImageModule(nn.Module):
    """Compresses (and reopens) a PIL image with a PIL codec (codec.py:79-111)."""

    def __init__(self, returns_file_size=False, open_kwargs=None, **save_kwargs):
        super().__init__()
        self.returns_file_size = returns_file_size
        self.open_kwargs = open_kwargs if isinstance(open_kwargs, dict) else dict()
        self.save_kwargs = save_kwargs

    def forward(self, pil_img, *args):
        decoded, n_bytes = _pil_codec_round_trip(pil_img, self.save_kwargs, self.open_kwargs)
        return (decoded, n_bytes) if self.returns_file_size else decoded

    def __repr__(self):
        return self.__class__.__name__ + '(returns_file_size={}, open_kwargs={}, save_kwargs={})'.format(
            self.returns_file_size, self.open_kwargs, self.save_kwargs)


@register_codec_transform_module
class PILTensorModule(nn.Module):
    """Compresses (and reconstructs) a tensor [C,H,W] with a PIL codec, three channels per image (codec.py:114-186)."""

    def __init__(self, returns_file_size=False, open_kwargs=None, **save_kwargs):
        super().__init__()
        self.returns_file_size = returns_file_size
        self.open_kwargs = open_kwargs if isinstance(open_kwargs, dict) else dict()
        self.save_kwargs = save_kwargs

    @staticmethod
    def _channel_groups(x):
        """[C,H,W] -> groups of three channels (RGB images); a remainder of one goes as a grey image, a remainder of two as
        two grey images (a two-channel tensor has no PIL mode)."""
        groups = list(x.split(3, dim=0))
        if groups[-1].shape[0] == 2:
            groups[-1:] = list(groups[-1].split(1, dim=0))
        return groups

    def forward(self, x, *args):
        # written against the behaviour of codec.py:141-186: per channel group, scale with the group's own extrema AS THE
        # REFERENCE DOES -- (v - min) / max on the way in, v * max + min on the way out, which is not a min-max scaling and
        # must not be "fixed"  --, 8-bit image through the codec, back to a tensor on x's device.  The
        # payload is the coded images plus the two pickled lists of 0-dim tensors the receiver needs to undo the scaling.
        lows, highs, restored, coded_bytes = [], [], [], 0
        for group in self._channel_groups(x):
            high, low = group.max(), group.min()
            decoded, n_bytes = _pil_codec_round_trip(to_pil_image((group - low) / high), self.save_kwargs, self.open_kwargs)
            coded_bytes += n_bytes
            if group.shape[0] == 1 and decoded.mode != 'L':     # a grey image some codecs reopen as RGB
                decoded = decoded.convert('L')
            restored.append(to_tensor(decoded).to(x.device) * high + low)
            lows.append(low)
            highs.append(high)
        features = torch.vstack(restored)
        if not self.returns_file_size:
            return features
        side_info = get_binary_object_size(lows, unit_size=1) + get_binary_object_size(highs, unit_size=1)
        return features, coded_bytes + side_info

    def __repr__(self):
        return self.__class__.__name__ + '(returns_file_size={}, open_kwargs={}, save_kwargs={})'.format(
            self.returns_file_size, self.open_kwargs, self.save_kwargs)


def get_transform(name):
    for d in (CODEC_TRANSFORM_MODULE_DICT, MISC_TRANSFORM_MODULE_DICT, TORCHVISION_TRANSFORM_DICT):
        if name in d:
            return d[name]
    return None
