"""Learned image-compression models for the neural INPUT compression baseline of the reference
(configs/ilsvrc2012/input_compression/factorized_prior-resnet50.yaml:60-65: `bmshj2018_factorized`, quality 8,
built through sc2bench/models/registry.py:58-105 from CompressAI's zoo and driven by
`NeuralInputCompressionClassifier`, sc2bench/models/wrapper.py:80-135).

`FactorizedPrior` keeps CompressAI's architecture, parameter / buffer names and API (`forward`, `compress`,
`decompress`, `update`, `aux_loss`, `load_state_dict`): g_a = 4 x Conv(k5, s2, p2, bias) with 3 GDN in between,
g_s = 4 x ConvTranspose(k5, s2, p2, output_padding 1, bias) with 3 inverse GDN, an `EntropyBottleneck(M)` on the latent.
Every tensor-sized computation runs in the HIP library: the convolutions / transposed convolutions on the implicit-GEMM
MFMA kernel (bias in the epilogue), GDN as a 1x1 GEMM on x^2 with rsqrt / sqrt fused in the epilogue, the entropy
bottleneck and the range coder as in the supervised-compression bottleneck.
"""
import torch
from torch import nn

from . import hip
from .entropy import CompressionModel, GDN, HipConv2d, HipConvTranspose2d, _require_device

COMPRESSION_MODEL_CLASS_DICT = dict()
COMPRESSION_MODEL_FUNC_DICT = dict()


def register_compression_model_class(cls):
    COMPRESSION_MODEL_CLASS_DICT[cls.__name__] = cls
    return cls


def register_compression_model_func(func):
    COMPRESSION_MODEL_FUNC_DICT[func.__name__] = func
    return func


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return HipConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return HipConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              output_padding=stride - 1, padding=kernel_size // 2)


def _run_transform(seq, x_nhwc, last_out_format):
    h = x_nhwc
    n = len(seq)
    for i, m in enumerate(seq):
        fmt = last_out_format if i == n - 1 else hip.OUT_BF16_NHWC
        h = m.forward_nhwc(h, out_format=fmt)
    return h


@register_compression_model_class
class FactorizedPrior(CompressionModel):
    """Factorized-prior model of Balle et al. 2018 as packaged by CompressAI (`compressai.models.FactorizedPrior`).

    :param N: channels of the transforms
    :param M: channels of the latent
    """

    def __init__(self, N, M, **kwargs):
        super().__init__(entropy_bottleneck_channels=M)
        self.g_a = nn.Sequential(
            conv(3, N), GDN(N),
            conv(N, N), GDN(N),
            conv(N, N), GDN(N),
            conv(N, M),
        )
        self.g_s = nn.Sequential(
            deconv(M, N), GDN(N, inverse=True),
            deconv(N, N), GDN(N, inverse=True),
            deconv(N, N), GDN(N, inverse=True),
            deconv(N, 3),
        )
        self.N = N
        self.M = M
        for prefix, seq in (('g_a', self.g_a), ('g_s', self.g_s)):
            for i, mod in enumerate(seq):
                mod._tag = '{}.{}'.format(prefix, i)

    @property
    def downsampling_factor(self):
        return 2 ** 4

    # ---- transforms on the device ---------------------------------------------------------------- #
    def analysis(self, x):
        """g_a(x): f32 NCHW image batch -> f32 NCHW latent."""
        _require_device(x, 'FactorizedPrior')
        x_nhwc = hip.nchw_f32_to_nhwc_bf16(x.float().contiguous(), 8)
        return _run_transform(self.g_a, x_nhwc, hip.OUT_F32_NCHW)

    def synthesis_nhwc(self, y_hat_nhwc):
        """g_s on a bf16 NHWC latent -> f32 NCHW reconstruction."""
        out = _run_transform(self.g_s, y_hat_nhwc, hip.OUT_F32_NHWC)
        return out.permute(0, 3, 1, 2).contiguous()

    def synthesis(self, y_hat):
        _require_device(y_hat, 'FactorizedPrior')
        return self.synthesis_nhwc(hip.nchw_f32_to_nhwc_bf16(y_hat.float().contiguous(), y_hat.shape[1]))

    # ---- CompressAI API ---------------------------------------------------------------------------- #
    def forward(self, x):
        y = self.analysis(x)
        y_hat, y_likelihoods = self.entropy_bottleneck(y)
        x_hat = self.synthesis(y_hat)
        return {'x_hat': x_hat, 'likelihoods': {'y': y_likelihoods}}

    def compress(self, x):
        y = self.analysis(x)
        y_strings = self.entropy_bottleneck.compress(y)
        return {'strings': [y_strings], 'shape': y.size()[-2:]}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 1
        _, y_hat_nhwc = self.entropy_bottleneck.decompress_to_device(strings[0], tuple(shape), want_f32=False, want_nhwc=True)
        x_hat = self.synthesis_nhwc(y_hat_nhwc).clamp_(0, 1)
        return {'x_hat': x_hat}

    # ---- compress -> decompress in stages (pipeline.StagePipeline through NeuralInputCompressionClassifier)
    stage_front_takes_out = False

    def stage_front(self, x, out=None):
        y = self.analysis(x)
        return self.entropy_bottleneck.symbols_device(y), tuple(y.shape[-2:])

    def stage_coder(self, sym, hw_shape, dequantized=False):
        eb = self.entropy_bottleneck
        hw = hw_shape[0] * hw_shape[1]
        buf, off, nb, st = eb.encode_symbols_device(sym, hw)
        if dequantized:
            y_hat = eb.decode_dequantize_device(buf, off, nb, sym.shape[1], hw_shape)
            if y_hat is not None:
                return y_hat, nb, st
        return eb.decode_symbols_device(buf, off, nb, sym.shape[1], hw), nb, st

    def stage_back(self, decoded, hw_shape):
        y_hat = decoded if decoded.dtype == torch.bfloat16 else self.entropy_bottleneck.dequantize_device(decoded, hw_shape)[1]
        return self.synthesis_nhwc(y_hat).clamp_(0, 1)

    @classmethod
    def from_state_dict(cls, state_dict):
        N = state_dict['g_a.0.weight'].size(0)
        M = state_dict['g_a.6.weight'].size(0)
        net = cls(N, M)
        net.load_state_dict(state_dict)
        return net


# compressai.zoo.image: quality -> (N, M) of bmshj2018-factorized
FACTORIZED_CFGS = {1: (128, 192), 2: (128, 192), 3: (128, 192), 4: (128, 192), 5: (128, 192),
                   6: (192, 320), 7: (192, 320), 8: (192, 320)}


@register_compression_model_func
def bmshj2018_factorized(quality, metric='mse', pretrained=False, progress=True, **kwargs):
    """compressai.zoo.bmshj2018_factorized.  Pretrained weights are downloaded upstream; offline they are read from
    $SC2_PRETRAINED_DIR/bmshj2018-factorized-{metric}-{quality}.pth (a CompressAI state dict, old `_matrix{i}` key names
    accepted) -- if that file is absent the model keeps its random initialisation and says so."""
    import logging
    import os
    import warnings
    if metric not in ('mse', 'ms-ssim'):
        raise ValueError('Invalid metric "{}"'.format(metric))
    if quality < 1 or quality > 8:
        raise ValueError('Invalid quality "{}", should be between (1, 8)'.format(quality))
    model = FactorizedPrior(*FACTORIZED_CFGS[quality], **kwargs)
    if pretrained:
        root = os.environ.get('SC2_PRETRAINED_DIR')
        path = os.path.join(root, 'bmshj2018-factorized-{}-{}.pth'.format(metric, quality)) if root else None
        if path and os.path.isfile(path):
            from .ckpt import _torch_load
            model.load_state_dict(_torch_load(path))
        else:
            msg = ('bmshj2018_factorized(quality={}, pretrained=True): no local weights ({}); the model is RANDOMLY '
                   'INITIALISED'.format(quality, path or 'SC2_PRETRAINED_DIR unset'))
            if os.environ.get('SC2_STRICT_WEIGHTS') == '1':
                raise FileNotFoundError(msg)
            warnings.warn(msg)
            logging.getLogger(__name__).warning(msg)
    return model


def get_compression_model(compression_model_config, device):
    """sc2bench/models/registry.py:83-105: {'key', 'kwargs', 'src_ckpt'?, 'update'?} -> model on `device`, CDF tables
    built unless `update: False`."""
    if compression_model_config is None:
        return None
    name = compression_model_config['key']
    kwargs = compression_model_config.get('kwargs') or dict()
    ckpt_path = compression_model_config.get('src_ckpt', None)
    if name in COMPRESSION_MODEL_FUNC_DICT or name in COMPRESSION_MODEL_CLASS_DICT:
        builder = COMPRESSION_MODEL_FUNC_DICT.get(name) or COMPRESSION_MODEL_CLASS_DICT[name]
        model = builder(**kwargs)
        if ckpt_path is not None:
            from .ckpt import load_ckpt
            load_ckpt(ckpt_path, model=model, strict=None)
        if compression_model_config.get('update', True):
            model.update()
        return model.to(device)
    raise ValueError('compression_model_name `{}` is not expected'.format(name))
