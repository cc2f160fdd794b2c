"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the "next" rows of SURVEY.md 8(f) ranks 3 and 4:

* ``GDN`` (squared form) and ``FactorizedPrior`` / ``bmshj2018_factorized`` -- CompressAI 1.2.x
  ``compressai/layers/gdn.py``, ``compressai/models/google.py`` (FactorizedPrior: g_a = conv k5 s2 x4 with 3 GDN,
  g_s = deconv k5 s2 output_padding 1 x4 with 3 inverse GDN, EntropyBottleneck(M); compress / decompress with the
  clamp to [0, 1]) and ``compressai/zoo/image.py`` (quality -> (N, M)); the reference reaches them at
  sc2bench/models/registry.py:58-105 and drives them from sc2bench/models/wrapper.py:80-135.
* ``AdaptivePad`` -- sc2bench/transforms/misc.py:105-154 (including its 'equal_side' test, not the docstring's 'hw').
* ``pil_tensor_module`` -- sc2bench/transforms/codec.py:114-186 (``PILTensorModule.forward``): channel groups of 3
  (a trailing group of 2 is split into 1 + 1), ``(x - min) / max`` normalisation (sic, :159) and ``* max + min``
  (:170), torchvision's ``to_pil_image`` (float -> ``mul(255).byte()``) / ``to_tensor`` (uint8 / 255), JPEG through
  PIL, file size = encoded bytes + pickled sizes of the two Python lists of 0-d tensors (:173-177).
* ``neural_input_compression_forward`` -- wrapper.py:121-135; ``codec_feature_compression_forward`` -- wrapper.py:178-193
  (``torch.hstack`` of the per-sample results, as the reference has it).

PARITY UNPINNED for the same reason as cpu_ref.py (CompressAI / torchvision / torchdistill are not installable here and
the reference holds no vectors).  The JPEG bytes depend on the Pillow / libjpeg build: fixtures record both versions.
"""
import pickle
import sys
from io import BytesIO

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .cpu_ref import EntropyBottleneck, NonNegativeParametrizer


class GDN(nn.Module):
    """y = x / sqrt(beta + gamma * x^2)  (inverse: x * sqrt(beta + gamma * x^2))."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        _, C, _, _ = x.size()
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(C, C, 1, 1)
        norm = F.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm


def _conv(i, o, k=5, s=2):
    return nn.Conv2d(i, o, kernel_size=k, stride=s, padding=k // 2)


def _deconv(i, o, k=5, s=2):
    return nn.ConvTranspose2d(i, o, kernel_size=k, stride=s, output_padding=s - 1, padding=k // 2)


class FactorizedPrior(nn.Module):
    def __init__(self, N, M):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(M)
        self.g_a = nn.Sequential(_conv(3, N), GDN(N), _conv(N, N), GDN(N), _conv(N, N), GDN(N), _conv(N, M))
        self.g_s = nn.Sequential(_deconv(M, N), GDN(N, inverse=True), _deconv(N, N), GDN(N, inverse=True),
                                 _deconv(N, N), GDN(N, inverse=True), _deconv(N, 3))
        self.N, self.M = N, M

    def forward(self, x, noise=None):
        y = self.g_a(x)
        y_hat, y_likelihoods = self.entropy_bottleneck(y, noise=noise)
        return {'x_hat': self.g_s(y_hat), 'likelihoods': {'y': y_likelihoods}}

    def update(self, force=False):
        return self.entropy_bottleneck.update(force=force)

    def aux_loss(self):
        return self.entropy_bottleneck.loss()

    def compress(self, x):
        y = self.g_a(x)
        return {'strings': [self.entropy_bottleneck.compress(y)], 'shape': y.size()[-2:]}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 1
        y_hat = self.entropy_bottleneck.decompress(strings[0], shape)
        return {'x_hat': self.g_s(y_hat).clamp_(0, 1)}


FACTORIZED_CFGS = {1: (128, 192), 2: (128, 192), 3: (128, 192), 4: (128, 192), 5: (128, 192),
                   6: (192, 320), 7: (192, 320), 8: (192, 320)}


def bmshj2018_factorized(quality, metric='mse'):
    return FactorizedPrior(*FACTORIZED_CFGS[quality])


# --------------------------------------------------------------------------- #
# transforms
# --------------------------------------------------------------------------- #
def adaptive_pad(x, fill=0, padding_position='hw', padding_mode='constant', factor=128):
    """x: tensor [..., H, W] -> right/bottom padded (or 'equal_side': left/top AND right/bottom by half) tensor."""
    height, width = x.shape[-2:]
    v = 0 if height % factor == 0 else int((height // factor + 1) * factor - height)
    h = 0 if width % factor == 0 else int((width // factor + 1) * factor - width)
    assert (v + height) % factor == 0 and (h + width) % factor == 0
    if padding_position == 'equal_side':       # torchvision pad([l/r, t/b]): the same amount on both sides
        return F.pad(x, [h // 2, h // 2, v // 2, v // 2], mode=padding_mode, value=fill)
    return F.pad(x, [0, h, 0, v], mode=padding_mode, value=fill)


def to_pil_image(t):
    """torchvision.transforms.functional.to_pil_image for a float or uint8 CHW tensor with 1 or 3 channels."""
    from PIL import Image
    if t.is_floating_point():
        t = t.mul(255).byte()
    arr = np.transpose(t.cpu().numpy(), (1, 2, 0))
    if arr.shape[2] == 1:
        return Image.fromarray(arr[:, :, 0], mode='L')
    assert arr.shape[2] == 3
    return Image.fromarray(arr, mode='RGB')


def to_tensor(pil_img):
    """torchvision.transforms.functional.to_tensor for 8-bit PIL images: HWC uint8 -> CHW float / 255."""
    arr = np.array(pil_img, copy=True)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def binary_object_size(obj, unit_size=1):
    return sys.getsizeof(pickle.dumps(obj)) / unit_size


def pil_tensor_module(x, open_kwargs=None, **save_kwargs):
    """-> (reconstructed tensor [C,H,W], file size in bytes)."""
    from PIL import Image
    open_kwargs = open_kwargs or dict()
    groups = list(x.split(3, dim=0))
    if groups[-1].shape[0] == 2:
        groups = groups[:-1] + list(groups[-1].split(1, dim=0))
    file_size = 0
    maxs, mins, recon = [], [], []
    for g in groups:
        mx, mn = g.max(), g.min()
        maxs.append(mx)
        mins.append(mn)
        img = to_pil_image((g - mn) / mx)
        buf = BytesIO()
        img.save(buf, **save_kwargs)
        file_size += buf.tell()
        img = Image.open(buf, **open_kwargs)
        if g.shape[0] == 1 and img.mode != 'L':
            img = img.convert('L')
        recon.append(to_tensor(img) * mx + mn)
    out = torch.vstack(recon)
    file_size += binary_object_size(mins) + binary_object_size(maxs)
    return out, file_size


def codec_feature_compression_forward(encoder, codec, decoder, classifier, x):
    """-> (logits, [file sizes])."""
    z = encoder(x)
    sizes, parts = [], []
    for sub in z:
        sub, size = codec(sub)
        sizes.append(size)
        parts.append(sub.unsqueeze(0))
    z = torch.hstack(parts)
    z = decoder(z)
    return classifier(torch.flatten(z, 1)), sizes


def neural_input_compression_forward(pre_transform, compression_model, post_transform, classifier, x):
    """-> (logits, compressed object)."""
    if pre_transform is not None:
        x = pre_transform(x)
    obj = compression_model.compress(x)
    x = compression_model.decompress(**obj)
    if isinstance(x, dict):
        x = x['x_hat']
    if post_transform is not None:
        x = post_transform(x)
    return classifier(x), obj
