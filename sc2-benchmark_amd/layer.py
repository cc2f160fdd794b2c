"""Bottleneck layers: host-side mirror of sc2bench/models/layer.py for the hot path.

Same registry (`register_layer_class`, `LAYER_CLASS_DICT`, `get_layer`: layer.py:11-38, 820-835), same
class names, constructor arguments, module paths (`encoder`, `decoder`, `entropy_bottleneck`), methods
(`forward/encode/decode/update/aux_loss/_get_means/load_state_dict`) and mode switching
(layer.py:535-550) as the reference, so `bottleneck_config: {key: 'FPBasedResNetBottleneck', ...}` from
the reference's YAML files builds this class unchanged.  The arithmetic runs in libsc2amd.so: activations
stay bf16 NHWC between the implicit-GEMM kernels; GDN1 is a 1x1 MFMA GEMM with fused epilogue.
"""
import torch
from torch import nn

from . import hip
from .entropy import CompressionModel, GDN1, HipConv2d, _require_device

LAYER_CLASS_DICT = dict()
LAYER_FUNC_DICT = dict()


def register_layer_class(cls):
    """Registers a layer class under its class name (layer.py:15-25)."""
    LAYER_CLASS_DICT[cls.__name__] = cls
    return cls


def register_layer_func(func):
    """Registers a function that builds a layer module (layer.py:28-38)."""
    LAYER_FUNC_DICT[func.__name__] = func
    return func


class EntropyBottleneckLayer(CompressionModel):
    """An entropy bottleneck as a stand-alone layer (layer.py:346-398)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.updated = False

    def forward(self, x):
        return self.entropy_bottleneck(x)

    def compress(self, x):
        strings = self.entropy_bottleneck.compress(x)
        return {'strings': [strings], 'shape': x.size()[-2:]}

    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 1
        return self.entropy_bottleneck.decompress(strings[0], shape)

    def update(self, force=False):
        self.updated = True
        return super().update(force=force)


class BaseBottleneck(CompressionModel):
    """Abstract entropy-bottleneck-based layer (layer.py:401-441)."""

    def __init__(self, entropy_bottleneck_channels):
        super().__init__(entropy_bottleneck_channels=entropy_bottleneck_channels)
        self.updated = False

    def encode(self, *args, **kwargs):
        raise NotImplementedError()

    def decode(self, *args, **kwargs):
        raise NotImplementedError()

    def forward(self, *args):
        raise NotImplementedError()

    def update(self, force=False):
        self.updated = True
        return super().update(force=force)


@register_layer_class
class FPBasedResNetBottleneck(BaseBottleneck):
    """Factorized-prior encoder / entropy bottleneck / decoder for ResNet (layer.py:444-550).

    :param num_input_channels: number of input channels
    :param num_bottleneck_channels: number of bottleneck (latent) channels
    :param num_target_channels: number of output channels of the decoder
    :param encoder_channel_sizes: 4 channel counts of the encoder or None
    :param decoder_channel_sizes: 4 channel counts of the decoder or None

    ``output_format``: 'f32_nchw' (what the reference returns; default) or 'bf16_nhwc' (a bf16
    channels_last view, zero-copy input for a bf16 task head).
    """

    def __init__(self, num_input_channels=3, num_bottleneck_channels=24, num_target_channels=256,
                 encoder_channel_sizes=None, decoder_channel_sizes=None):
        if encoder_channel_sizes is None:
            encoder_channel_sizes = \
                [num_input_channels, num_bottleneck_channels * 4, num_bottleneck_channels * 2, num_bottleneck_channels]
        if decoder_channel_sizes is None:
            decoder_channel_sizes = \
                [encoder_channel_sizes[-1], num_target_channels * 2, num_target_channels, num_target_channels]
        super().__init__(entropy_bottleneck_channels=num_bottleneck_channels)
        e, d = encoder_channel_sizes, decoder_channel_sizes
        self.encoder = nn.Sequential(
            HipConv2d(e[0], e[1], kernel_size=5, stride=2, padding=2, bias=False),
            GDN1(e[1]),
            HipConv2d(e[1], e[2], kernel_size=5, stride=2, padding=2, bias=False),
            GDN1(e[2]),
            HipConv2d(e[2], e[3], kernel_size=2, stride=1, padding=0, bias=False)
        )
        self.decoder = nn.Sequential(
            HipConv2d(d[0], d[1], kernel_size=2, stride=1, padding=1, bias=False),
            GDN1(d[1], inverse=True),
            HipConv2d(d[1], d[2], kernel_size=2, stride=1, padding=0, bias=False),
            GDN1(d[2], inverse=True),
            HipConv2d(d[2], d[3], kernel_size=2, stride=1, padding=1, bias=False)
        )
        for prefix, seq in (('enc', self.encoder), ('dec', self.decoder)):
            for i, mod in enumerate(seq):
                mod._tag = '{}.{}{}'.format(prefix, 'igdn' if getattr(mod, 'inverse', False) else
                                            'gdn' if isinstance(mod, GDN1) else 'conv', i)
        self.output_format = 'f32_nchw'
        self.fuse_gdn = True    # conv + GDN1 in one launch where one tile holds all output channels (encoder)
        self._conv0_pack = None
        self._conv0_key = None

    # ---- fused pipelines on bf16 NHWC ---------------------------------------------------------- #
    def _conv0_packed(self):
        w = self.encoder[0].weight
        key = (w._version, w.device, w.data_ptr())
        if self._conv0_key != key:
            self._conv0_pack = hip.pack_conv0_weight_pairs(w)
            self._conv0_key = key
        return self._conv0_pack

    def _uses_pair_conv0(self, x):
        c0 = self.encoder[0]
        return (c0.in_channels <= 4 and c0.kernel_size == (5, 5) and c0.stride == (2, 2) and c0.padding == (2, 2)
                and x.shape[-1] % 2 == 0 and c0.out_channels % 8 == 0)

    def analysis(self, x):
        """encoder(x): f32 NCHW image batch -> f32 NCHW latent (layer.py:475-483)."""
        _require_device(x, 'FPBasedResNetBottleneck')
        c0, g1, c2, g3, c4 = self.encoder
        x = x.float()
        fuse0 = self.fuse_gdn and c0.out_channels in hip.FUSABLE_GDN_CHANNELS
        fuse2 = self.fuse_gdn and c2.out_channels in hip.FUSABLE_GDN_CHANNELS
        if self._uses_pair_conv0(x):
            N, _, H, W = x.shape
            x4 = hip.nchw_f32_to_nhwc_bf16(x, 4)                      # [N,H,W,4]
            xp = x4.view(N, H, W // 2, 8)                             # pixel pairs
            if fuse0:
                beta, gamma = g1.effective()
                h = hip.conv2d_fwd(xp, self._conv0_packed(), c0.out_channels, 5, 3, (2, 1), (2, 1),
                                   epilogue=hip.EPI_FUSED_IGDN if g1.inverse else hip.EPI_FUSED_GDN, ep_x=gamma,
                                   ep_beta=beta, tag=c0._tag + '+' + g1._tag)
            else:
                h = hip.conv2d_fwd(xp, self._conv0_packed(), c0.out_channels, 5, 3, (2, 1), (2, 1), tag=c0._tag)
        else:
            fuse0 = False
            cin = c0.in_channels
            xin = hip.nchw_f32_to_nhwc_bf16(x, (cin + 7) // 8 * 8)
            if cin % 8 != 0:
                w = torch.zeros((c0.out_channels, xin.shape[-1]) + tuple(c0.kernel_size), dtype=c0.weight.dtype,
                                device=c0.weight.device)
                w[:, :cin] = c0.weight.detach()
                h = hip.conv2d_fwd(xin, hip.pack_conv_weight(w), c0.out_channels, c0.kernel_size[0],
                                   c0.kernel_size[1], c0.stride, c0.padding)
            else:
                h = c0.forward_nhwc(xin)
        if not fuse0:
            h = g1.forward_nhwc(h)
        if fuse2:
            beta, gamma = g3.effective()
            h = hip.conv2d_fwd(h, c2.packed_weight(), c2.out_channels, c2.kernel_size[0], c2.kernel_size[1], c2.stride,
                               c2.padding, epilogue=hip.EPI_FUSED_IGDN if g3.inverse else hip.EPI_FUSED_GDN,
                               ep_x=gamma, ep_beta=beta, tag=c2._tag + '+' + g3._tag, k_order=c2.k_order())
        else:
            h = g3.forward_nhwc(c2.forward_nhwc(h))
        return c4.forward_nhwc(h, out_format=hip.OUT_F32_NCHW)

    def synthesis_nhwc(self, y_hat_nhwc):
        """decoder on a bf16 NHWC latent (layer.py:485-493); output per ``self.output_format``."""
        c0, g1, c2, g3, c4 = self.decoder
        if (self.fuse_gdn and g1.in_channels == c0.out_channels and
                hip.conv2x2_gdn512_supported(c0.in_channels, c0.out_channels, c0.kernel_size[0], c0.kernel_size[1],
                                             c0.stride, c0.padding)):
            beta, gamma = g1.effective_fragments()   # conv + (inverse) GDN1(512) in one persistent launch
            h = hip.conv2x2_gdn512_fwd(y_hat_nhwc, c0.packed_weight(hip.K_TAP_MAJOR), gamma, beta, g1.inverse, tag=c0._tag + '+' + g1._tag)
        else:
            h = c0.forward_nhwc(y_hat_nhwc)
            h = g1.forward_nhwc(h)
        if self.fuse_gdn and hip.conv_fused_gdn_supported(tuple(h.shape), c2.out_channels, c2.kernel_size[0],
                                                          c2.kernel_size[1], c2.stride, c2.padding):
            beta, gamma = g3.effective()   # conv + inverse GDN1 in one launch (256-wide big tile holds all channels)
            h = hip.conv2d_fwd(h, c2.packed_weight(), c2.out_channels, c2.kernel_size[0], c2.kernel_size[1], c2.stride,
                               c2.padding, epilogue=hip.EPI_FUSED_IGDN if g3.inverse else hip.EPI_FUSED_GDN,
                               ep_x=gamma, ep_beta=beta, tag=c2._tag + '+' + g3._tag, k_order=c2.k_order())
        else:
            h = c2.forward_nhwc(h)
            h = g3.forward_nhwc(h)
        if self.output_format == 'bf16_nhwc':
            out = c4.forward_nhwc(h, out_format=hip.OUT_BF16_NHWC)
            return out.permute(0, 3, 1, 2)  # logical NCHW, channels_last memory
        return c4.forward_nhwc(h, out_format=hip.OUT_F32_NCHW)

    def synthesis(self, y_hat):
        """decoder(y_hat) for an f32 NCHW latent."""
        _require_device(y_hat, 'FPBasedResNetBottleneck')
        return self.synthesis_nhwc(hip.nchw_f32_to_nhwc_bf16(y_hat.float(), y_hat.shape[1]))

    # ---- reference API -------------------------------------------------------------------------- #
    def encode(self, x, **kwargs):
        """-> {'strings': [list of N byte strings], 'shape': latent spatial size} (layer.py:496-507)."""
        latent = self.analysis(x)
        latent_strings = self.entropy_bottleneck.compress(latent)
        return {'strings': [latent_strings], 'shape': latent.size()[-2:]}

    def decode(self, strings, shape):
        """strings, shape -> decoder output (layer.py:509-521)."""
        eb = self.entropy_bottleneck
        dev = eb._quantized_cdf.device
        if dev.type != 'cuda':
            raise hip.Sc2Error('FPBasedResNetBottleneck.decode: module is on {}; HIP device required'.format(dev))
        buf, off, nb = eb.pack_strings(strings[0], dev)
        _, y_hat_nhwc = eb.decompress_device(buf, off, nb, tuple(shape), want_f32=False, want_nhwc=True)
        return self.synthesis_nhwc(y_hat_nhwc)

    def encode_device(self, x):
        """Device-resident encode: (buf, offset, nbytes, status, latent spatial size); no host sync."""
        latent = self.analysis(x)
        buf, off, nb, st = self.entropy_bottleneck.compress_device(latent)
        return buf, off, nb, st, tuple(latent.shape[-2:])

    def decode_device(self, buf, off, nb, shape):
        _, y_hat_nhwc = self.entropy_bottleneck.decompress_device(buf, off, nb, tuple(shape), want_f32=False,
                                                                  want_nhwc=True)
        return self.synthesis_nhwc(y_hat_nhwc)

    def _get_means(self, x):
        medians = self.entropy_bottleneck._get_medians().detach()
        spatial_dims = len(x.size()) - 2
        medians = self.entropy_bottleneck._extend_ndims(medians, spatial_dims)
        return medians.expand(x.size(0), *([-1] * (spatial_dims + 1)))

    def _needs_grad(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _forward2train(self, x):
        if self._needs_grad(x):
            from .autograd import bottleneck_forward2train_autograd
            return bottleneck_forward2train_autograd(self, x)
        encoded_obj = self.analysis(x)
        y_hat, y_likelihoods = self.entropy_bottleneck(encoded_obj)
        return self.synthesis(y_hat)

    def forward(self, x):
        # if fine-tune or evaluate after "update"
        if self.updated:
            if not self.training:
                encoded_obj = self.encode(x)
                decoded_obj = self.decode(**encoded_obj)
                return decoded_obj
            if self._needs_grad(x):
                from .autograd import bottleneck_forward_updated_autograd
                return bottleneck_forward_updated_autograd(self, x)
            encoded_output = self.analysis(x)
            decoder_input = self.entropy_bottleneck.dequantize(
                self.entropy_bottleneck.quantize(encoded_output, 'dequantize', self._get_means(encoded_output)))
            decoder_input = decoder_input.detach()
            return self.synthesis(decoder_input)
        return self._forward2train(x)


def get_layer(cls_or_func_name, **kwargs):
    """Gets a layer module by registered class or function name (layer.py:820-835)."""
    if cls_or_func_name in LAYER_CLASS_DICT:
        return LAYER_CLASS_DICT[cls_or_func_name](**kwargs)
    elif cls_or_func_name in LAYER_FUNC_DICT:
        return LAYER_FUNC_DICT[cls_or_func_name](**kwargs)
    return None
