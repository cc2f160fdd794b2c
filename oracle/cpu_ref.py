"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (plain PyTorch fp32/fp64 ops, in the order the reference would
execute them) of the floating-point half of the hot path:

* ``LowerBound``, ``NonNegativeParametrizer``, ``GDN1``  -- CompressAI
  ``compressai/layers/gdn.py`` + ``compressai/ops/parametrizers.py``, used by the
  reference at sc2bench/models/layer.py:478,481,488,491.
* ``EntropyBottleneck`` (forward / likelihood / quantize / dequantize / update /
  loss / compress / decompress) -- CompressAI 1.2.x
  ``compressai/entropy_models/entropy_models.py``; reference call sites
  layer.py:506,520,524-526,531,545-547.
* ``FPBasedResNetBottleneck`` -- follows sc2bench/models/layer.py:444-550 line by
  line in behaviour (structure 464-494, encode 496-507, decode 509-521,
  _get_means 523-527, _forward2train 529-533, forward 535-550).
* ``GaussianConditional``, ``get_scale_table`` -- CompressAI 1.2.x entropy_models.py / models/google.py;
  ``SHPBasedResNetBottleneck`` / ``MSHPBasedResNetBottleneck`` -- sc2bench/models/layer.py:553-817.
* ``SplittableResNet`` forward order -- sc2bench/models/backbone.py:225-258.
* ``BppLoss`` -- sc2bench/loss.py:20-37.  ``file_size`` -- sc2bench/analysis.py:126-134
  (torchdistill ``get_binary_object_size`` = ``sys.getsizeof(pickle.dumps(obj))/unit``).

The integer half (CDF quantisation, rANS) is in ``oracle/rans_oracle.c``.

PARITY UNPINNED: CompressAI / torchdistill / torchvision are third-party wheels
that are not under /root/reference, not installed and not installable here
(no network); the reference has no tests or golden vectors.
The torch CPU ops used below (conv2d, softplus, tanh, sigmoid, round) are the
genuine arithmetic the reference would have dispatched to on a CPU device, so
for the FP rows this is the strongest oracle available; fixtures generated from
it live in tests/golden/ together with the generating script.
"""
import math
import pickle
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import rans as _rans


# --------------------------------------------------------------------------- #
# CompressAI ops restated
# --------------------------------------------------------------------------- #
class _LowerBoundFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, grad_output):
        x, bound = ctx.saved_tensors
        pass_through_if = (x >= bound) | (grad_output < 0)
        return pass_through_if * grad_output, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer('bound', torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFunction.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer('pedestal', torch.Tensor([pedestal]))
        bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self.lower_bound = LowerBound(bound)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        out = out ** 2 - self.pedestal
        return out


class GDN1(nn.Module):
    """y = x / (beta + gamma * |x|)  (inverse: x * (beta + gamma * |x|))."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        beta = torch.ones(in_channels)
        self.beta = nn.Parameter(self.beta_reparam.init(beta))
        self.gamma_reparam = NonNegativeParametrizer()
        gamma = float(gamma_init) * torch.eye(in_channels)
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma))

    def forward(self, x):
        _, C, _, _ = x.size()
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma)
        gamma = gamma.reshape(C, C, 1, 1)
        norm = F.conv2d(torch.abs(x), gamma, beta)
        if not self.inverse:
            norm = 1.0 / norm
        return x * norm


class EntropyBottleneck(nn.Module):
    """Factorised-prior entropy model (CompressAI 1.2.x semantics)."""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3),
                 likelihood_bound=1e-9, entropy_coder_precision=16):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer('_offset', torch.IntTensor())
        self.register_buffer('_quantized_cdf', torch.IntTensor())
        self.register_buffer('_cdf_length', torch.IntTensor())

        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        channels = self.channels
        self.matrices = nn.ParameterList()
        self.biases = nn.ParameterList()
        self.factors = nn.ParameterList()
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            matrix = torch.Tensor(channels, filters[i + 1], filters[i])
            matrix.data.fill_(init)
            self.matrices.append(nn.Parameter(matrix))
            bias = torch.Tensor(channels, filters[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.biases.append(nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, filters[i + 1], 1)
                nn.init.zeros_(factor)
                self.factors.append(nn.Parameter(factor))

        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer('target', torch.Tensor([-target, 0, target]))

    # -- helpers ----------------------------------------------------------- #
    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    @staticmethod
    def _extend_ndims(tensor, n):
        return tensor.reshape(-1, *([1] * n)) if n > 0 else tensor.reshape(-1)

    @staticmethod
    def _build_indexes(size):
        dims = len(size)
        C = size[1]
        view_dims = np.ones((dims,), dtype=np.int64)
        view_dims[1] = -1
        indexes = torch.arange(C).view(*view_dims)
        indexes = indexes.int()
        return indexes.repeat(size[0], 1, *size[2:])

    def _logits_cumulative(self, inputs, stop_gradient):
        logits = inputs
        for i in range(len(self.filters) + 1):
            matrix = self.matrices[i]
            if stop_gradient:
                matrix = matrix.detach()
            logits = torch.matmul(F.softplus(matrix), logits)
            bias = self.biases[i]
            if stop_gradient:
                bias = bias.detach()
            logits = logits + bias
            if i < len(self.filters):
                factor = self.factors[i]
                if stop_gradient:
                    factor = factor.detach()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def _likelihood(self, inputs, stop_gradient=False):
        half = float(0.5)
        lower = self._logits_cumulative(inputs - half, stop_gradient=stop_gradient)
        upper = self._logits_cumulative(inputs + half, stop_gradient=stop_gradient)
        likelihood = torch.sigmoid(upper) - torch.sigmoid(lower)
        return likelihood, lower, upper

    # -- quantisation ------------------------------------------------------ #
    def quantize(self, inputs, mode, means=None, noise=None):
        if mode not in ('noise', 'dequantize', 'symbols'):
            raise ValueError('Invalid quantization mode: "{}"'.format(mode))
        if mode == 'noise':
            if noise is None:
                half = float(0.5)
                noise = torch.empty_like(inputs).uniform_(-half, half)
            return inputs + noise
        outputs = inputs.clone()
        if means is not None:
            outputs -= means
        outputs = torch.round(outputs)
        if mode == 'dequantize':
            if means is not None:
                outputs += means
            return outputs
        return outputs.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.type(dtype)
        return outputs

    # -- forward ----------------------------------------------------------- #
    def forward(self, x, training=None, noise=None):
        """``noise`` (same shape as x) replaces the U(-.5,.5) draw so tests are deterministic."""
        if training is None:
            training = self.training
        perm = [1, 0] + list(range(2, x.ndim))
        x = x.permute(*perm).contiguous()
        shape = x.size()
        values = x.reshape(x.size(0), 1, -1)
        if noise is not None:
            noise = noise.permute(*perm).contiguous().reshape(x.size(0), 1, -1)
        outputs = self.quantize(values, 'noise' if training else 'dequantize', self._get_medians(), noise=noise)
        likelihood, _, _ = self._likelihood(outputs)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        outputs = outputs.reshape(shape).permute(*perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*perm).contiguous()
        return outputs, likelihood

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    # -- CDF tables -------------------------------------------------------- #
    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        medians = self.quantiles[:, 0, 1]
        minima = medians - self.quantiles[:, 0, 0]
        minima = torch.ceil(minima).int()
        minima = torch.clamp(minima, min=0)
        maxima = self.quantiles[:, 0, 2] - medians
        maxima = torch.ceil(maxima).int()
        maxima = torch.clamp(maxima, min=0)
        self._offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max().item()
        samples = torch.arange(max_length)
        samples = samples[None, :] + pmf_start[:, None, None]
        pmf, lower, upper = self._likelihood(samples, stop_gradient=True)
        pmf = pmf[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._cdf_length = pmf_length + 2
        return True

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = _rans.pmf_to_quantized_cdf(prob.detach().tolist(), self.entropy_coder_precision)
            cdf[i, :_cdf.size] = torch.from_numpy(_cdf.astype(np.int64)).int()
        return cdf

    def _check_tables(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError('Uninitialized CDFs. Run update() first')
        if self._offset.numel() == 0:
            raise ValueError('Uninitialized offsets. Run update() first')
        if self._cdf_length.numel() == 0:
            raise ValueError('Uninitialized CDF lengths. Run update() first')

    def symbols(self, x):
        """round(x - median).int() in NCHW order (the integers handed to the range coder)."""
        medians = self._extend_ndims(self._get_medians().detach(), x.ndim - 2)
        medians = medians.expand(x.size(0), *([-1] * (x.ndim - 1)))
        return self.quantize(x, 'symbols', medians)

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        symbols = self.symbols(x)
        self._check_tables()
        strings = []
        for i in range(symbols.size(0)):
            strings.append(_rans.encode_with_indexes(
                symbols[i].reshape(-1).numpy(), indexes[i].reshape(-1).numpy(),
                self._quantized_cdf.numpy(), self._cdf_length.reshape(-1).numpy(),
                self._offset.reshape(-1).numpy()))
        return strings

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), *size)
        indexes = self._build_indexes(output_size)
        medians = self._extend_ndims(self._get_medians().detach(), len(size))
        medians = medians.expand(len(strings), *([-1] * (len(size) + 1)))
        self._check_tables()
        outputs = torch.empty(indexes.size(), dtype=torch.int32)
        for i, s in enumerate(strings):
            values = _rans.decode_with_indexes(
                s, indexes[i].reshape(-1).numpy(), self._quantized_cdf.numpy(),
                self._cdf_length.reshape(-1).numpy(), self._offset.reshape(-1).numpy())
            outputs[i] = torch.from_numpy(values).reshape(outputs[i].size())
        return self.dequantize(outputs, medians, medians.dtype)


# --------------------------------------------------------------------------- #
# sc2bench layer restated (layer.py:401-550)
# --------------------------------------------------------------------------- #
class FPBasedResNetBottleneck(nn.Module):
    def __init__(self, num_input_channels=3, num_bottleneck_channels=24, num_target_channels=256,
                 encoder_channel_sizes=None, decoder_channel_sizes=None):
        super().__init__()
        if encoder_channel_sizes is None:
            encoder_channel_sizes = [num_input_channels, num_bottleneck_channels * 4,
                                     num_bottleneck_channels * 2, num_bottleneck_channels]
        if decoder_channel_sizes is None:
            decoder_channel_sizes = [encoder_channel_sizes[-1], num_target_channels * 2,
                                     num_target_channels, num_target_channels]
        self.entropy_bottleneck = EntropyBottleneck(num_bottleneck_channels)
        self.updated = False
        e, d = encoder_channel_sizes, decoder_channel_sizes
        self.encoder = nn.Sequential(
            nn.Conv2d(e[0], e[1], kernel_size=5, stride=2, padding=2, bias=False),
            GDN1(e[1]),
            nn.Conv2d(e[1], e[2], kernel_size=5, stride=2, padding=2, bias=False),
            GDN1(e[2]),
            nn.Conv2d(e[2], e[3], kernel_size=2, stride=1, padding=0, bias=False)
        )
        self.decoder = nn.Sequential(
            nn.Conv2d(d[0], d[1], kernel_size=2, stride=1, padding=1, bias=False),
            GDN1(d[1], inverse=True),
            nn.Conv2d(d[1], d[2], kernel_size=2, stride=1, padding=0, bias=False),
            GDN1(d[2], inverse=True),
            nn.Conv2d(d[2], d[3], kernel_size=2, stride=1, padding=1, bias=False)
        )

    def update(self, force=False):
        self.updated = True
        return self.entropy_bottleneck.update(force=force)

    def aux_loss(self):
        return self.entropy_bottleneck.loss()

    def encode(self, x, **kwargs):
        latent = self.encoder(x)
        latent_strings = self.entropy_bottleneck.compress(latent)
        return {'strings': [latent_strings], 'shape': latent.size()[-2:]}

    def decode(self, strings, shape):
        latent_hat = self.entropy_bottleneck.decompress(strings[0], shape)
        return self.decoder(latent_hat)

    def _get_means(self, x):
        medians = self.entropy_bottleneck._get_medians().detach()
        spatial_dims = len(x.size()) - 2
        medians = self.entropy_bottleneck._extend_ndims(medians, spatial_dims)
        return medians.expand(x.size(0), *([-1] * (spatial_dims + 1)))

    def _forward2train(self, x, noise=None):
        encoded_obj = self.encoder(x)
        y_hat, y_likelihoods = self.entropy_bottleneck(encoded_obj, noise=noise)
        return self.decoder(y_hat)

    def forward(self, x, noise=None):
        if self.updated:
            if not self.training:
                return self.decode(**self.encode(x))
            encoded_output = self.encoder(x)
            decoder_input = self.entropy_bottleneck.dequantize(
                self.entropy_bottleneck.quantize(encoded_output, 'dequantize', self._get_means(encoded_output)))
            decoder_input = decoder_input.detach()
            return self.decoder(decoder_input)
        return self._forward2train(x, noise=noise)


# --------------------------------------------------------------------------- #
# GaussianConditional + hyperprior bottlenecks (layer.py:553-817)
# --------------------------------------------------------------------------- #
SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """compressai.models.google.get_scale_table (imported by the reference at layer.py:5, used at :700)."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class GaussianConditional(nn.Module):
    """CompressAI 1.2.x ``GaussianConditional`` (entropy_models.py) as the reference uses it: constructed with
    ``GaussianConditional(None)`` (layer.py:627), ``build_indexes`` / ``compress(y, indexes, means=)`` /
    ``decompress(strings, indexes, dtype | means=)`` (layer.py:646-647,665,776,785), ``forward(y, scales, means=)``
    (:679,794), ``quantize`` / ``dequantize`` (:691-693,811-813), ``update_scale_table`` (:702)."""

    def __init__(self, scale_table=None, scale_bound=0.11, tail_mass=1e-9, likelihood_bound=1e-9,
                 entropy_coder_precision=16):
        super().__init__()
        if scale_table is not None and (len(scale_table) < 1 or list(scale_table) != sorted(scale_table) or
                                        any(s <= 0 for s in scale_table)):
            raise ValueError('Invalid scale_table "({})"'.format(scale_table))
        self.tail_mass = float(tail_mass)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        if scale_bound is None and scale_table:
            scale_bound = scale_table[0]
        if scale_bound <= 0:
            raise ValueError('Invalid parameters')
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer('_offset', torch.IntTensor())
        self.register_buffer('_quantized_cdf', torch.IntTensor())
        self.register_buffer('_cdf_length', torch.IntTensor())
        self.register_buffer('scale_table', self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer('scale_bound', torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(inputs):
        half = float(0.5)
        const = float(-(2 ** -0.5))
        return half * torch.erfc(const * inputs)

    @staticmethod
    def _standardized_quantile(quantile):
        import scipy.stats
        return scipy.stats.norm.ppf(quantile)

    quantize = EntropyBottleneck.quantize
    dequantize = staticmethod(EntropyBottleneck.dequantize)
    _pmf_to_cdf = EntropyBottleneck._pmf_to_cdf
    _check_tables = EntropyBottleneck._check_tables

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    def update(self):
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None])
        samples_scale = self.scale_table.unsqueeze(1)
        samples = samples.float()
        samples_scale = samples_scale.float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def _likelihood(self, inputs, scales, means=None):
        half = float(0.5)
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        upper = self._standardized_cumulative((half - values) / scales)
        lower = self._standardized_cumulative((-half - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None, training=None, noise=None):
        if training is None:
            training = self.training
        outputs = self.quantize(inputs, 'noise' if training else 'dequantize', means, noise=noise)
        likelihood = self._likelihood(outputs, scales, means)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes

    def compress(self, inputs, indexes, means=None):
        symbols = self.quantize(inputs, 'symbols', means)
        if len(inputs.size()) < 2:
            raise ValueError('Invalid `inputs` size. Expected a tensor with at least 2 dimensions.')
        if inputs.size() != indexes.size():
            raise ValueError('`inputs` and `indexes` should have the same size.')
        self._check_tables()
        strings = []
        for i in range(symbols.size(0)):
            strings.append(_rans.encode_with_indexes(
                symbols[i].reshape(-1).int().numpy(), indexes[i].reshape(-1).int().numpy(),
                self._quantized_cdf.numpy(), self._cdf_length.reshape(-1).int().numpy(),
                self._offset.reshape(-1).int().numpy()))
        return strings

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        if not isinstance(strings, (tuple, list)):
            raise ValueError('Invalid `strings` parameter type.')
        if not len(strings) == indexes.size(0):
            raise ValueError('Invalid strings or indexes parameters')
        self._check_tables()
        if means is not None and means.size()[:2] != indexes.size()[:2]:
            raise ValueError('Invalid means or indexes parameters')
        outputs = torch.empty(indexes.size(), dtype=torch.int32)
        for i, s in enumerate(strings):
            values = _rans.decode_with_indexes(
                s, indexes[i].reshape(-1).int().numpy(), self._quantized_cdf.numpy(),
                self._cdf_length.reshape(-1).int().numpy(), self._offset.reshape(-1).int().numpy())
            outputs[i] = torch.from_numpy(values).reshape(outputs[i].size())
        return self.dequantize(outputs, means, dtype)


class SHPBasedResNetBottleneck(nn.Module):
    """sc2bench/models/layer.py:553-720 restated (scale hyperprior)."""

    def __init__(self, num_input_channels=3, num_latent_channels=16, num_bottleneck_channels=24,
                 num_target_channels=256, h_a=None, h_s=None, g_a_channel_sizes=None, g_s_channel_sizes=None):
        super().__init__()
        if g_a_channel_sizes is None:
            g_a_channel_sizes = [num_input_channels, num_bottleneck_channels * 4, num_bottleneck_channels * 2,
                                 num_bottleneck_channels]
        else:
            num_bottleneck_channels = g_a_channel_sizes[3]
        if g_s_channel_sizes is None:
            g_s_channel_sizes = [g_a_channel_sizes[-1], num_target_channels * 2, num_target_channels,
                                 num_target_channels]
        self.entropy_bottleneck = EntropyBottleneck(num_latent_channels)
        self.updated = False
        a, g = g_a_channel_sizes, g_s_channel_sizes
        self.g_a = nn.Sequential(
            nn.Conv2d(a[0], a[1], kernel_size=5, stride=2, padding=2, bias=False), GDN1(a[1]),
            nn.Conv2d(a[1], a[2], kernel_size=5, stride=2, padding=2, bias=False), GDN1(a[2]),
            nn.Conv2d(a[2], a[3], kernel_size=2, stride=1, padding=0, bias=False))
        self.g_s = nn.Sequential(
            nn.Conv2d(g[0], g[1], kernel_size=2, stride=1, padding=1, bias=False), GDN1(g[1], inverse=True),
            nn.Conv2d(g[1], g[2], kernel_size=2, stride=1, padding=0, bias=False), GDN1(g[2], inverse=True),
            nn.Conv2d(g[2], g[3], kernel_size=2, stride=1, padding=1, bias=False))
        L, B = num_latent_channels, num_bottleneck_channels
        self.h_a = nn.Sequential(
            nn.Conv2d(B, L, kernel_size=5, stride=2, padding=1, bias=False), nn.ReLU(inplace=True),
            nn.Conv2d(L, L, kernel_size=5, stride=2, padding=2, bias=False)) if h_a is None else h_a
        self.h_s = nn.Sequential(
            nn.ConvTranspose2d(L, L, kernel_size=5, stride=2, padding=1, bias=False), nn.LeakyReLU(inplace=True),
            nn.ConvTranspose2d(L, L, kernel_size=5, stride=2, padding=1, bias=False), nn.LeakyReLU(inplace=True),
            nn.Conv2d(L, B, kernel_size=5, stride=1, padding=0, bias=False)) if h_s is None else h_s
        self.gaussian_conditional = GaussianConditional(None)
        self.num_latent_channels = num_latent_channels
        self.num_bottleneck_channels = num_bottleneck_channels

    def aux_loss(self):
        return self.entropy_bottleneck.loss()

    def encode(self, x, **kwargs):
        y = self.g_a(x)
        z = self.h_a(torch.abs(y))
        z_shape = z.size()[-2:]
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z_shape)
        scales_hat = self.h_s(z_hat)
        indices = self.gaussian_conditional.build_indexes(scales_hat)
        y_strings = self.gaussian_conditional.compress(y, indices)
        return {'strings': [y_strings, z_strings], 'shape': z_shape}

    def decode(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        scales_hat = self.h_s(z_hat)
        indices = self.gaussian_conditional.build_indexes(scales_hat)
        y_hat = self.gaussian_conditional.decompress(strings[0], indices, z_hat.dtype)
        return self.g_s(y_hat)

    def _get_means(self, x):
        medians = self.entropy_bottleneck._get_medians().detach()
        spatial_dims = len(x.size()) - 2
        medians = self.entropy_bottleneck._extend_ndims(medians, spatial_dims)
        return medians.expand(x.size(0), *([-1] * (spatial_dims + 1)))

    def _forward2train(self, x, noise_z=None, noise_y=None):
        y = self.g_a(x)
        z = self.h_a(torch.abs(y))
        z_hat, z_likelihoods = self.entropy_bottleneck(z, noise=noise_z)
        scales_hat = self.h_s(z_hat)
        y_hat, y_likelihoods = self.gaussian_conditional(y, scales_hat, noise=noise_y)
        self.last_likelihoods = (y_likelihoods, z_likelihoods)
        return self.g_s(y_hat)

    def forward(self, x, **kw):
        if self.updated:
            if not self.training:
                return self.decode(**self.encode(x))
            y = self.g_a(x)
            y_hat = self.gaussian_conditional.dequantize(
                self.gaussian_conditional.quantize(y, 'dequantize', self._get_means(y)))
            y_hat = y_hat.detach()
            return self.g_s(y_hat)
        return self._forward2train(x, **kw)

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        self.updated = True
        return updated


class MSHPBasedResNetBottleneck(SHPBasedResNetBottleneck):
    """sc2bench/models/layer.py:723-817 restated (mean-scale hyperprior)."""

    def __init__(self, num_input_channels=3, num_latent_channels=16, num_bottleneck_channels=24,
                 num_target_channels=256, g_a_channel_sizes=None, g_s_channel_sizes=None):
        L, B = num_latent_channels, num_bottleneck_channels
        h_a = nn.Sequential(
            nn.Conv2d(B, L, kernel_size=5, stride=2, padding=1, bias=False), nn.LeakyReLU(inplace=True),
            nn.Conv2d(L, L, kernel_size=5, stride=2, padding=2, bias=False))
        h_s = nn.Sequential(
            nn.ConvTranspose2d(L, L, kernel_size=5, stride=2, padding=1, bias=False), nn.LeakyReLU(inplace=True),
            nn.ConvTranspose2d(L, L * 3 // 2, kernel_size=5, stride=2, padding=1, bias=False),
            nn.LeakyReLU(inplace=True),
            nn.Conv2d(L * 3 // 2, B * 2, kernel_size=5, stride=1, padding=0, bias=False))
        super().__init__(num_input_channels=num_input_channels, num_latent_channels=num_latent_channels,
                         num_bottleneck_channels=num_bottleneck_channels, num_target_channels=num_target_channels,
                         h_a=h_a, h_s=h_s, g_a_channel_sizes=g_a_channel_sizes, g_s_channel_sizes=g_s_channel_sizes)

    def encode(self, x, **kwargs):
        y = self.g_a(x)
        z = self.h_a(y)
        z_strings = self.entropy_bottleneck.compress(z)
        z_shape = z.size()[-2:]
        z_hat = self.entropy_bottleneck.decompress(z_strings, z_shape)
        gaussian_params = self.h_s(z_hat)
        scales_hat, means_hat = gaussian_params.chunk(2, 1)
        indices = self.gaussian_conditional.build_indexes(scales_hat)
        y_strings = self.gaussian_conditional.compress(y, indices, means=means_hat)
        return {'strings': [y_strings, z_strings], 'shape': z_shape}

    def decode(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        gaussian_params = self.h_s(z_hat)
        scales_hat, means_hat = gaussian_params.chunk(2, 1)
        indices = self.gaussian_conditional.build_indexes(scales_hat)
        y_hat = self.gaussian_conditional.decompress(strings[0], indices, means=means_hat)
        return self.g_s(y_hat)

    def _forward2train(self, x, noise_z=None, noise_y=None):
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_likelihoods = self.entropy_bottleneck(z, noise=noise_z)
        gaussian_params = self.h_s(z_hat)
        scales_hat, means_hat = gaussian_params.chunk(2, 1)
        y_hat, y_likelihoods = self.gaussian_conditional(y, scales_hat, means=means_hat, noise=noise_y)
        self.last_likelihoods = (y_likelihoods, z_likelihoods)
        return self.g_s(y_hat)

    def forward(self, x, **kw):
        if self.updated:
            if not self.training:
                return self.decode(**self.encode(x))
            y = self.g_a(x)
            z = self.h_a(y)
            z_hat = self.entropy_bottleneck.dequantize(
                self.entropy_bottleneck.quantize(z, 'dequantize', self._get_means(z)))
            gaussian_params = self.h_s(z_hat)
            scales_hat, means_hat = gaussian_params.chunk(2, 1)
            y_hat = self.gaussian_conditional.dequantize(
                self.gaussian_conditional.quantize(y, 'dequantize', means_hat))
            y_hat = y_hat.detach()
            return self.g_s(y_hat)
        return self._forward2train(x, **kw)


# --------------------------------------------------------------------------- #
# ResNet-50 tail (torchvision architecture, restated) + SplittableResNet order
# --------------------------------------------------------------------------- #
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride):
    downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
    layers = [_Bottleneck(inplanes, planes, stride, downsample)]
    for _ in range(1, blocks):
        layers.append(_Bottleneck(planes * 4, planes))
    return nn.Sequential(*layers)


class SplittableResNet50(nn.Module):
    """bottleneck_layer -> layer2 -> layer3 -> layer4 -> avgpool -> flatten -> fc (backbone.py:225-254)."""

    def __init__(self, bottleneck_layer, num_classes=1000):
        super().__init__()
        self.bottleneck_layer = bottleneck_layer
        self.layer2 = _make_layer(256, 128, 4, 2)
        self.layer3 = _make_layer(512, 256, 6, 2)
        self.layer4 = _make_layer(1024, 512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        self.bottleneck_updated = False
        self.last_encoded = None

    def update(self):
        self.bottleneck_layer.update()
        self.bottleneck_updated = True

    def forward(self, x):
        if self.bottleneck_updated and not self.training:
            enc = self.bottleneck_layer.encode(x)
            self.last_encoded = enc
            x = self.bottleneck_layer.decode(**enc)
        else:
            x = self.bottleneck_layer(x)
        x = self.layer4(self.layer3(self.layer2(x)))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


# --------------------------------------------------------------------------- #
# harness arithmetic
# --------------------------------------------------------------------------- #
def bpp_loss(features, likelihoods, reduction='mean'):
    """sc2bench/loss.py:28-37."""
    n, _, h, w = features.shape
    num_pixels = n * h * w
    if reduction == 'sum':
        return -likelihoods.log2().sum()
    if reduction == 'batchmean':
        return -likelihoods.log2().sum() / n
    return -likelihoods.log2().sum() / num_pixels


def file_size(compressed_obj, unit_size=1024):
    """torchdistill get_binary_object_size as used at sc2bench/analysis.py:133."""
    return sys.getsizeof(pickle.dumps(compressed_obj)) / unit_size


def perturb_quantiles(entropy_bottleneck):
    """Deterministic non-degenerate quantiles: [-(3+c%5), 0.25*(c%3), 4+c%7]."""
    with torch.no_grad():
        C = entropy_bottleneck.channels
        q = torch.zeros(C, 1, 3)
        for c in range(C):
            q[c, 0, 0] = -(3 + c % 5)
            q[c, 0, 1] = 0.25 * (c % 3)
            q[c, 0, 2] = 4 + c % 7
        entropy_bottleneck.quantiles.copy_(q)


def state_dict_fingerprint(module):
    out = OrderedDict()
    for k, v in module.state_dict().items():
        out[k] = (tuple(v.shape), str(v.dtype), float(v.double().sum()) if v.numel() else 0.0)
    return out


__all__ = ['LowerBound', 'NonNegativeParametrizer', 'GDN1', 'EntropyBottleneck', 'FPBasedResNetBottleneck',
           'SplittableResNet50', 'bpp_loss', 'file_size', 'perturb_quantiles', 'state_dict_fingerprint', 'math']
