"""Host-side mirror of the CompressAI classes the reference's bottleneck is built from.

sc2bench builds ``FPBasedResNetBottleneck`` out of ``compressai.layers.GDN1``,
``compressai.entropy_models.EntropyBottleneck`` and ``compressai.models.CompressionModel``
(sc2bench/models/layer.py:2-6, 401-494).  The classes here keep CompressAI's constructor arguments,
parameter / buffer names (so released checkpoints load), method names and error
behaviour, but every tensor-sized computation runs in the HIP library (``hip.py``); only
per-channel scalar work (reparametrisation of beta/gamma, softplus/tanh of the 58 bottleneck
parameters per channel, the once-per-model CDF table build) uses torch ops.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import hip


# --------------------------------------------------------------------------------------------- #
# compressai.ops restated on torch ops (parameter-sized tensors only)
# --------------------------------------------------------------------------------------------- #
class _LowerBoundFn(torch.autograd.Function):
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
    """max(x, bound) whose gradient passes when x >= bound or the gradient pushes x up."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer('bound', torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


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
        return out ** 2 - self.pedestal


def _require_device(x, what):
    if not x.is_cuda:
        raise hip.Sc2Error('{}: input is on {}; sc2bench_amd computes on a HIP device only (no CPU fallback)'
                           .format(what, x.device))


def _status_or(st):
    """OR over the per-stream status bit masks (the max of bit masks is not their OR: 1 and 2 give 2)."""
    if isinstance(st, torch.Tensor):
        st = st.detach().cpu().numpy()
    st = np.asarray(st)
    return int(np.bitwise_or.reduce(st.astype(np.int64).reshape(-1))) if st.size else 0


def _raise_on_status(st, what):
    """rANS status per stream: bit 0 = the row overflowed its stride, bit 1 = a symbol lies outside the codable range
    (|symbol - offset| >= 2^30: a non-finite or diverged latent; upstream's nibble loop never terminates there), bit 2 = a
    CDF-row index outside the table (host coder), bit 3 (decoders) = a corrupt, truncated or hostile stream: an escape that
    announces more than eight nibbles (upstream's decoder never returns from one made of 0xFF bytes) or words read past the
    end of the stream."""
    code = _status_or(st)
    if code & 2:
        raise ValueError('{}: a symbol is outside the codable range (non-finite or diverged latent)'.format(what))
    if code & 4:
        raise ValueError('{}: a CDF-row index lies outside the table'.format(what))
    if code & 8:
        raise ValueError('{}: a byte stream is corrupt or shorter than its symbols need'.format(what))
    if code:
        raise hip.Sc2Error('{}: rANS stream overflowed its maximum size'.format(what))


class HipConv2d(nn.Conv2d):
    """nn.Conv2d(bias=False) parameter holder whose forward is the implicit-GEMM MFMA kernel.

    Module-by-module use (``encoder(x)``) converts f32 NCHW <-> bf16 NHWC around the kernel; the
    bottleneck's own forward keeps activations in bf16 NHWC between layers (``bottleneck.py``).
    """

    def k_order(self):
        return hip.preferred_k_order(self.in_channels, self.kernel_size[0], self.kernel_size[1])

    def packed_weight(self, k_order=None):
        """Packed bf16 weights in this layer's preferred k order (or the one asked for); cached per parameter version."""
        order = self.k_order() if k_order is None else k_order
        key = (self.weight._version, self.weight.device, self.weight.data_ptr(), order)
        cache = self.__dict__.setdefault('_packed_cache', {})
        if cache.get('key') != key[:3]:
            cache.clear()
            cache['key'] = key[:3]
        if order not in cache:
            cache[order] = hip.pack_conv_weight(self.weight, order)
        return cache[order]

    def bias_f32(self):
        return None if self.bias is None else self.bias.detach().float().contiguous()

    def padded_weight(self, cin_pad):
        """Packed weights with the input channels zero-padded to `cin_pad` (a 3-channel image enters the kernels as
        8-channel NHWC); cached per parameter version."""
        key = (self.weight._version, self.weight.device, self.weight.data_ptr(), cin_pad)
        if getattr(self, '_padw_key', None) != key:
            w = torch.zeros((self.out_channels, cin_pad) + tuple(self.kernel_size), dtype=self.weight.dtype,
                            device=self.weight.device)
            w[:, :self.in_channels] = self.weight.detach()
            self._padw = hip.pack_conv_weight(w)
            self._padw_key = key
        return self._padw

    def forward_nhwc(self, x_nhwc, out_format=hip.OUT_BF16_NHWC):
        """x bf16 NHWC (channels possibly zero-padded beyond in_channels to a multiple of 8); a bias rides in the epilogue."""
        assert self.groups == 1 and self.dilation == (1, 1)
        epi, beta = (hip.EPI_NONE, None) if self.bias is None else (hip.EPI_BIAS, self.bias_f32())
        if x_nhwc.shape[-1] != self.in_channels:
            return hip.conv2d_fwd(x_nhwc, self.padded_weight(x_nhwc.shape[-1]), self.out_channels, self.kernel_size[0],
                                  self.kernel_size[1], self.stride, self.padding, epilogue=epi, ep_beta=beta,
                                  out_format=out_format, tag=getattr(self, '_tag', None))
        return hip.conv2d_fwd(x_nhwc, self.packed_weight(), self.out_channels, self.kernel_size[0],
                              self.kernel_size[1], self.stride, self.padding, epilogue=epi, ep_beta=beta,
                              out_format=out_format, tag=getattr(self, '_tag', None), k_order=self.k_order())

    def forward(self, x):
        _require_device(x, 'HipConv2d')
        x_nhwc = hip.nchw_f32_to_nhwc_bf16(x.float(), (self.in_channels + 7) // 8 * 8)
        return self.forward_nhwc(x_nhwc, out_format=hip.OUT_F32_NCHW)


class HipConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d parameter holder (h_s of the hyperprior bottlenecks, layer.py:612-621; with bias and
    output_padding, the `deconv` of CompressAI's bmshj2018_factorized synthesis transform) whose
    forward runs on the implicit-GEMM kernel: a transposed convolution is, per stride-parity class of the output, a
    stride-1 correlation with the flipped sub-filter of the taps that reach that class; each class is one launch that
    scatters its rows to every s-th output pixel (the data-gradient path of the training code, `hip.conv2d_dgrad`).
    The packed sub-filters are cached per parameter version."""

    def _cout_pad(self):
        return (self.out_channels + 7) // 8 * 8     # the kernels write whole 16-byte channel runs

    def _classes(self):
        key = (self.weight._version, self.weight.device, self.weight.data_ptr())
        if getattr(self, '_cls_key', None) != key:
            assert self.groups == 1 and self.dilation == (1, 1)
            cin, cout, KH, KW = self.weight.shape          # ConvTranspose2d weight: [in, out, kh, kw]
            sh, sw = self.stride
            ph, pw = self.padding
            wt = self.weight.detach().permute(1, 0, 2, 3)   # [out, in, kh, kw]: rows = output channels
            if self._cout_pad() != cout:
                wt = torch.cat([wt, wt.new_zeros((self._cout_pad() - cout,) + tuple(wt.shape[1:]))])
            classes = []
            for ch in range(sh):
                rh = (ch + ph) % sh
                khs = list(range(rh, KH, sh))
                for cw in range(sw):
                    rw = (cw + pw) % sw
                    kws = list(range(rw, KW, sw))
                    if not khs or not kws:
                        raise hip.Sc2Error('HipConvTranspose2d: a stride-parity class without taps (k={} s={} p={})'
                                           .format((KH, KW), (sh, sw), (ph, pw)))
                    qh, qw = (ch + ph - rh) // sh, (cw + pw - rw) // sw
                    pad_h, pad_w = len(khs) - 1 - qh, len(kws) - 1 - qw
                    if pad_h < 0 or pad_w < 0:
                        raise hip.Sc2Error('HipConvTranspose2d: unsupported geometry k={} s={} p={}'
                                           .format((KH, KW), (sh, sw), (ph, pw)))
                    sub = wt[:, :, khs][:, :, :, kws].flip(2, 3).contiguous()
                    classes.append((ch, cw, len(khs), len(kws), pad_h, pad_w, hip.pack_conv_weight(sub)))
            self._cls = classes
            self._cls_key = key
        return self._cls

    def forward_nhwc(self, x_nhwc, epilogue=hip.EPI_NONE, ep_beta=None, out_format=hip.OUT_BF16_NHWC):
        """x bf16 [N,H,W,Cin] -> [N,(H-1)s-2p+k+op,(W-1)s-2p+k+op,Cout] (bf16, or f32 NHWC)."""
        N, H, W, _ = x_nhwc.shape
        sh, sw = self.stride
        OH = (H - 1) * sh - 2 * self.padding[0] + self.kernel_size[0] + self.output_padding[0]
        OW = (W - 1) * sw - 2 * self.padding[1] + self.kernel_size[1] + self.output_padding[1]
        cout, cpad = self.out_channels, self._cout_pad()
        if self.bias is not None:      # the bias rides in each parity class's epilogue
            if epilogue != hip.EPI_NONE:
                raise hip.Sc2Error('HipConvTranspose2d: a bias and a fused activation epilogue together are not supported')
            epilogue, ep_beta = hip.EPI_BIAS, self.bias.detach().float().contiguous()
        if ep_beta is not None and cpad != cout:
            ep_beta = torch.cat([ep_beta, ep_beta.new_zeros(cpad - cout)])
        out = torch.empty((N, OH, OW, cpad), dtype=torch.bfloat16 if out_format == hip.OUT_BF16_NHWC else torch.float32,
                          device=x_nhwc.device)
        for ch, cw, nkh, nkw, pad_h, pad_w, packed in self._classes():
            rows = (OH - ch + sh - 1) // sh if OH > ch else 0
            cols = (OW - cw + sw - 1) // sw if OW > cw else 0
            if rows == 0 or cols == 0:
                continue
            if sh == 1 and sw == 1:
                hip.conv2d_fwd(x_nhwc, packed, cpad, nkh, nkw, 1, (pad_h, pad_w), epilogue=epilogue, ep_beta=ep_beta,
                               out_format=out_format, out=out, tag=getattr(self, '_tag', None))
            else:
                hip.conv2d_fwd(x_nhwc, packed, cpad, nkh, nkw, 1, (pad_h, pad_w), epilogue=epilogue, ep_beta=ep_beta,
                               out_format=out_format, tag=getattr(self, '_tag', None),
                               scatter=(rows, cols, out, sh, sw, ch, cw))
        return out if cpad == cout else out[..., :cout]

    def forward(self, x, output_size=None):
        _require_device(x, 'HipConvTranspose2d')
        out = self.forward_nhwc(hip.nchw_f32_to_nhwc_bf16(x.float()), out_format=hip.OUT_F32_NHWC)
        return out.permute(0, 3, 1, 2).contiguous()


def run_hip_sequence(seq, x_nhwc, a_op=hip.AOP_NONE, last_out_format=hip.OUT_F32_NCHW):
    """Runs an nn.Sequential of HipConv2d / HipConvTranspose2d / ReLU / LeakyReLU (the h_a / h_s transforms of the
    hyperprior bottlenecks) on bf16 NHWC activations; an activation is fused into the preceding conv's epilogue.
    `a_op` applies to the first conv's input (|y| of the scale hyperprior).  Any other module runs as a torch
    module on the device, on an f32 NCHW copy.  Returns the last layer's output in `last_out_format`."""
    mods = list(seq)
    h = x_nhwc
    i = 0
    first = True
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        last = (i + 1 >= len(mods)) or (isinstance(nxt, (nn.ReLU, nn.LeakyReLU)) and i + 2 >= len(mods))
        fmt = last_out_format if last else hip.OUT_BF16_NHWC
        if isinstance(m, (HipConv2d, HipConvTranspose2d)):
            epi, beta = hip.EPI_NONE, None
            if isinstance(nxt, nn.ReLU):
                epi = hip.EPI_BIAS_RELU
            elif isinstance(nxt, nn.LeakyReLU) and abs(nxt.negative_slope - 0.01) < 1e-12:
                epi = hip.EPI_BIAS_LEAKY_RELU
            if epi != hip.EPI_NONE:
                beta = torch.zeros(m.out_channels, dtype=torch.float32, device=h.device)
                i += 1
            if isinstance(m, HipConv2d):
                assert m.bias is None
                h = hip.conv2d_fwd(h, m.packed_weight(), m.out_channels, m.kernel_size[0], m.kernel_size[1], m.stride,
                                   m.padding, a_op=a_op if first else hip.AOP_NONE, epilogue=epi, ep_beta=beta,
                                   out_format=fmt, tag=getattr(m, '_tag', None), k_order=m.k_order())
            else:
                if first and a_op != hip.AOP_NONE:
                    h = h.abs()
                if fmt == hip.OUT_F32_NCHW:
                    h = m.forward_nhwc(h, epi, beta, out_format=hip.OUT_F32_NHWC).permute(0, 3, 1, 2).contiguous()
                else:
                    h = m.forward_nhwc(h, epi, beta, out_format=fmt)
        else:
            x32 = h.float().permute(0, 3, 1, 2)
            if first and a_op != hip.AOP_NONE:
                x32 = x32.abs()
            y32 = m(x32.contiguous())
            h = y32.contiguous() if last and last_out_format == hip.OUT_F32_NCHW else \
                hip.nchw_f32_to_nhwc_bf16(y32.float().contiguous())
        first = False
        i += 1
    return h


class GDN1(nn.Module):
    """Simplified generalized divisive normalisation: y = x / (beta + gamma |x|); inverse: x * (...).

    Same parameters, initialisation and state-dict keys as compressai.layers.GDN1 (used by the
    reference at layer.py:478,481,488,491).  The channel reduction gamma |x| is a 1x1 implicit GEMM on
    the matrix cores with |.| applied on operand load and the divide / multiply fused in the epilogue.
    """

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        beta_min = float(beta_min)
        gamma_init = float(gamma_init)
        self.inverse = bool(inverse)
        self.in_channels = int(in_channels)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        beta = torch.ones(in_channels)
        beta = self.beta_reparam.init(beta)
        self.beta = nn.Parameter(beta)
        self.gamma_reparam = NonNegativeParametrizer()
        gamma = gamma_init * torch.eye(in_channels)
        gamma = self.gamma_reparam.init(gamma)
        self.gamma = nn.Parameter(gamma)

    def effective(self):
        """(beta f32 [C], packed bf16 gamma [Cpad, Kpad]) on the parameter device; cached per parameter version."""
        key = (self.beta._version, self.gamma._version, self.gamma.device, self.gamma.data_ptr())
        if getattr(self, '_eff_key', None) != key:
            with torch.no_grad():
                beta = self.beta_reparam(self.beta).float().contiguous()
                gamma = self.gamma_reparam(self.gamma)
                C = self.in_channels
                self._eff = (beta, hip.pack_conv_weight(gamma.reshape(C, C, 1, 1)))
            self._eff_key = key
        return self._eff

    def effective_fragments(self):
        """(beta f32 [C], gamma as bf16 MFMA-fragment-major blocks: hip.pack_gamma_fragments); cached like effective()."""
        key = (self.beta._version, self.gamma._version, self.gamma.device, self.gamma.data_ptr())
        if getattr(self, '_frag_key', None) != key:
            with torch.no_grad():
                beta = self.beta_reparam(self.beta).float().contiguous()
                self._frag = (beta, hip.pack_gamma_fragments(self.gamma_reparam(self.gamma)))
            self._frag_key = key
        return self._frag

    def forward_nhwc(self, x_nhwc, out_format=hip.OUT_BF16_NHWC):
        beta, gamma_packed = self.effective()
        return hip.conv2d_fwd(x_nhwc, gamma_packed, self.in_channels, 1, 1, 1, 0, a_op=hip.AOP_ABS,
                              epilogue=hip.EPI_IGDN if self.inverse else hip.EPI_GDN, out_format=out_format,
                              ep_x=x_nhwc, ep_beta=beta, tag=getattr(self, '_tag', None))

    def forward(self, x):
        _require_device(x, 'GDN1')
        x_nhwc = hip.nchw_f32_to_nhwc_bf16(x.float())
        return self.forward_nhwc(x_nhwc, out_format=hip.OUT_F32_NCHW)


class GDN(GDN1):
    """Generalized divisive normalisation, squared form: y = x / sqrt(beta + gamma x^2); inverse: x * sqrt(...).

    Same parameters, initialisation and state-dict keys as compressai.layers.GDN (the analysis / synthesis transforms
    of `bmshj2018_factorized`, which the reference builds at sc2bench/models/registry.py:73-80 for the neural input
    compression configs).  gamma x^2 is a 1x1 implicit GEMM on the matrix cores with the square applied to the
    operand fragments and rsqrt / sqrt fused in the epilogue."""

    def forward_nhwc(self, x_nhwc, out_format=hip.OUT_BF16_NHWC):
        beta, gamma_packed = self.effective()
        return hip.conv2d_fwd(x_nhwc, gamma_packed, self.in_channels, 1, 1, 1, 0, a_op=hip.AOP_SQUARE,
                              epilogue=hip.EPI_IGDN2 if self.inverse else hip.EPI_GDN2, out_format=out_format,
                              ep_x=x_nhwc, ep_beta=beta, tag=getattr(self, '_tag', None))


class _HostTablesMixin(object):
    """Lifetime of the prepared host-coder tables (`_host_tables_cache`, a ctypes handle kept in the instance __dict__):
    dropped whenever the integer tables can have changed -- update(), load_state_dict(), .to() / .cuda() / .float() -- and never
    pickled or deep-copied (a ctypes object holding pointers cannot be)."""

    def _invalidate_host_tables(self):
        self.__dict__['_tables_epoch'] = self.__dict__.get('_tables_epoch', 0) + 1
        self.__dict__.pop('_host_tables_cache', None)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate_host_tables()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate_host_tables()
        return out

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_host_tables_cache', None)
        return state

    def _host_tables(self):
        """Prepared tables of the library's host coder (csrc/rans_host.cpp), rebuilt when the integer tables changed."""
        # keyed on (i) an explicit epoch that update() / load_state_dict() / .to() bump: the buffers are REPLACED by fresh tensors
        # there (version 0 again, and the caching allocator hands the same address out), so pointer identity can hit a stale entry;
        # (ii) the buffers' versions and shapes: an in-place write (`_quantized_cdf.copy_(...)` in a tool or a test) changes the
        # tables without going through any of those entry points
        bufs = (self._quantized_cdf, self._offset, self._cdf_length)
        key = (self.__dict__.get('_tables_epoch', 0), tuple(t._version for t in bufs), tuple(tuple(t.shape) for t in bufs))
        cached = self.__dict__.get('_host_tables_cache')
        if cached is None or cached[0] != key:
            cdf, cdf_len, offset = self._tables()
            cached = (key, hip.HostRansTables(cdf, cdf_len, offset))
            self.__dict__['_host_tables_cache'] = cached
        return cached[1]


# --------------------------------------------------------------------------------------------- #
# EntropyBottleneck
# --------------------------------------------------------------------------------------------- #
class EntropyBottleneck(_HostTablesMixin, nn.Module):
    """Factorised-prior entropy model with CompressAI 1.2.x semantics.

    forward / quantize / dequantize / compress / decompress run in the HIP library; ``update()``
    builds the integer CDF tables once per model on the host (as the reference's C++ helper does).
    """

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3),
                 likelihood_bound=1e-9, entropy_coder_precision=16, **kwargs):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self.likelihood_bound = float(likelihood_bound)
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

    # ---- small helpers (same names as upstream; the reference calls the first two, layer.py:524-526)
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
        """Parameter-sized use only (aux loss on [C,1,3] quantiles, CDF grid in update())."""
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

    def _check_filters(self):
        if self.filters != (3, 3, 3, 3):
            raise hip.Sc2Error('the HIP entropy-bottleneck kernels are built for filters=(3,3,3,3) '
                               '(the only configuration sc2bench uses); got {}'.format(self.filters))

    def effective_params(self):
        """f32 [C, 64] block the kernels read (layout: include/sc2_bottleneck.h); differentiable."""
        self._check_filters()
        C = self.channels
        parts = []
        for i in range(5):
            parts.append(F.softplus(self.matrices[i]).reshape(C, -1))
            parts.append(self.biases[i].reshape(C, -1))
            if i < 4:
                parts.append(torch.tanh(self.factors[i]).reshape(C, -1))
        parts.append(self._get_medians().reshape(C, 1))
        p = torch.cat(parts, dim=1)
        return F.pad(p, (0, hip.EB_PARAM_STRIDE - p.shape[1])).float().contiguous()

    def _cached_params(self):
        ps = list(self.matrices) + list(self.biases) + list(self.factors) + [self.quantiles]
        key = tuple(p._version for p in ps) + (self.quantiles.device, self.quantiles.data_ptr())
        if getattr(self, '_eff_key', None) != key:
            with torch.no_grad():
                self._eff = self.effective_params()
            self._eff_key = key
        return self._eff

    # ---- quantisation (EntropyModel.quantize / dequantize; reference call layer.py:545-547)
    def quantize(self, inputs, mode, means=None):
        if mode not in ('noise', 'dequantize', 'symbols'):
            raise ValueError('Invalid quantization mode: "{}"'.format(mode))
        _require_device(inputs, 'EntropyBottleneck.quantize')
        if mode == 'noise':
            half = float(0.5)
            noise = torch.empty_like(inputs).uniform_(-half, half)
            return inputs + noise
        x = inputs.float().contiguous()
        C = x.shape[1]
        if means is None:
            med = torch.zeros(C, dtype=torch.float32, device=x.device)
        else:
            med = means.detach().float().reshape(means.shape[0], C, -1)[0, :, 0].contiguous()
        sym = hip.eb_symbols(x, med)
        if mode == 'symbols':
            return sym
        return hip.eb_dequantize(sym, med, want_f32=True)[0]

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs = outputs + means
        else:
            outputs = inputs.type(dtype)
        return outputs

    # ---- forward (reference call layer.py:531)
    def forward(self, x, training=None, noise=None):
        """Returns (y_hat, likelihoods), both f32 with x's shape.  ``noise`` overrides the U(-.5,.5) draw."""
        if training is None:
            training = self.training
        _require_device(x, 'EntropyBottleneck.forward')
        y = x.float().contiguous()
        if torch.is_grad_enabled() and (y.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import eb_forward_autograd
            return eb_forward_autograd(self, y, training, noise)
        params = self._cached_params()
        if training:
            if noise is None:
                half = float(0.5)
                noise = torch.empty_like(y).uniform_(-half, half)
            y_hat, _, lik, _ = hip.eb_forward(y, params, hip.EB_NOISE, noise=noise.float().contiguous(),
                                              lik_bound=self.likelihood_bound if self.use_likelihood_bound else 0.0)
        else:
            y_hat, _, lik, _ = hip.eb_forward(y, params, hip.EB_DEQUANTIZE,
                                              lik_bound=self.likelihood_bound if self.use_likelihood_bound else 0.0)
        return y_hat, lik

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    # ---- CDF tables (EntropyBottleneck.update; reached from layer.py:431-441)
    @torch.no_grad()
    def update(self, force=False, update_quantiles=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        # Host evaluation in f32 with torch CPU ops, in upstream's op order: the integer tables depend on
        # p*65536 rounding, so they are built where the reference's CPU path builds them.
        cpu = _CpuReplica(self)
        quantiles = self.quantiles.detach().cpu()
        medians = quantiles[:, 0, 1]
        minima = medians - quantiles[:, 0, 0]
        minima = torch.ceil(minima).int()
        minima = torch.clamp(minima, min=0)
        maxima = quantiles[:, 0, 2] - medians
        maxima = torch.ceil(maxima).int()
        maxima = torch.clamp(maxima, min=0)
        offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max().item()
        samples = torch.arange(max_length)
        samples = samples[None, :] + pmf_start[:, None, None]
        half = float(0.5)
        lower = cpu.logits_cumulative(samples - half)
        upper = cpu.logits_cumulative(samples + half)
        pmf = torch.sigmoid(upper) - torch.sigmoid(lower)
        pmf = pmf[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = offset.to(dev)
        self._quantized_cdf = quantized_cdf.to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        self._invalidate_host_tables()
        return True

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = hip.pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, :_cdf.size(0)] = _cdf
        return cdf

    def _check_cdf_size(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError('Uninitialized CDFs. Run update() first')
        if len(self._quantized_cdf.size()) != 2:
            raise ValueError('Invalid CDF size {}'.format(self._quantized_cdf.size()))

    def _check_offsets_size(self):
        if self._offset.numel() == 0:
            raise ValueError('Uninitialized offsets. Run update() first')
        if len(self._offset.size()) != 1:
            raise ValueError('Invalid offsets size {}'.format(self._offset.size()))

    def _check_cdf_length(self):
        if self._cdf_length.numel() == 0:
            raise ValueError('Uninitialized CDF lengths. Run update() first')
        if len(self._cdf_length.size()) != 1:
            raise ValueError('Invalid offsets size {}'.format(self._cdf_length.size()))

    def _tables(self):
        self._check_cdf_size()
        self._check_cdf_length()
        self._check_offsets_size()
        return (self._quantized_cdf.contiguous(), self._cdf_length.reshape(-1).int().contiguous(),
                self._offset.reshape(-1).int().contiguous())

    def _median_vector(self):
        return self._get_medians().detach().reshape(-1).float().contiguous()

    # ---- entropy coding on the device
    def compress_device(self, x):
        """x: f32 [N,C,*spatial] on device -> (buf u8 [N,stride], offset i32 [N], nbytes i32 [N]) on device.

        No host synchronisation unless a row overflowed its (generous) default stride.
        """
        if len(x.size()) < 2:
            raise ValueError('Invalid `inputs` size. Expected a tensor with at least 2 dimensions.')
        _require_device(x, 'EntropyBottleneck.compress')
        cdf, cdf_len, offset = self._tables()
        y = x.float().contiguous()
        N, C = y.shape[0], y.shape[1]
        if C != cdf.shape[0]:
            raise ValueError('`inputs` has {} channels but the CDF table has {} rows'.format(C, cdf.shape[0]))
        hw = y.numel() // (N * C)
        sym = hip.eb_symbols(y, self._median_vector())
        buf, off, nb, st = hip.rans_encode_batch(sym.view(N, C * hw), cdf, cdf_len, offset, index_div=hw)
        return buf, off, nb, st

    # ---- stage-wise device API (lets a caller put the serial coder on its own HIP stream)
    def symbols_device(self, x):
        """x: f32 [N,C,*spatial] -> int32 symbols [N, C*prod(spatial)] (round(x - median), NCHW order)."""
        _require_device(x, 'EntropyBottleneck.symbols')
        y = x.float().contiguous()
        N, C = y.shape[0], y.shape[1]
        return hip.eb_symbols(y, self._median_vector()).view(N, -1)

    def encode_symbols_device(self, sym, hw, out_stride=None):
        """sym: int32 [N, C*hw] -> (buf, offset, nbytes, status) on device."""
        cdf, cdf_len, offset = self._tables()
        return hip.rans_encode_batch(sym, cdf, cdf_len, offset, index_div=hw, out_stride=out_stride)

    def decode_symbols_device(self, buf, off, nb, n_sym, hw):
        """-> int32 symbols [N, n_sym] on device."""
        cdf, cdf_len, offset = self._tables()
        return hip.rans_decode_batch(buf, off, nb, n_sym, cdf, cdf_len, offset, index_div=hw)[0]

    def decode_dequantize_device(self, buf, off, nb, n_sym, size):
        """decompress + dequantize in one coder launch: -> y_hat bf16 NHWC [N, h, w, C] on device, or None when the tables
        do not fit the fused pass (the caller then uses decode_symbols_device + dequantize_device)."""
        cdf, cdf_len, offset = self._tables()
        hw = size[0] * size[1]
        if not hip.rans_decode_dequantize_supported(cdf.shape[0], cdf.shape[1]) or n_sym != cdf.shape[0] * hw:
            return None
        y_hat, _, _ = hip.rans_decode_dequantize_batch(buf, off, nb, n_sym, cdf, cdf_len, offset, hw, self._median_vector())
        return y_hat.view(y_hat.shape[0], size[0], size[1], cdf.shape[0])

    def dequantize_device(self, sym, size, want_f32=False, want_nhwc=True):
        """int32 symbols [N, C*prod(size)] -> (y_hat f32 NCHW or None, y_hat bf16 NHWC or None)."""
        C = self._quantized_cdf.shape[0]
        return hip.eb_dequantize(sym.view(sym.shape[0], C, *size), self._median_vector(), want_f32=want_f32,
                                 want_nhwc=want_nhwc)

    @staticmethod
    def _host_staging(staging, slot, N, n_sym):
        """the two pinned buffers of one host-coder slot (symbols out, decoded symbols in) + the event that guards their reuse;
        kept between calls: cudaHostAlloc of 2 x 74 MB costs tens of milliseconds"""
        key = (slot, N, n_sym)
        bufs = staging.get(key)
        if bufs is None:
            # (outside inference mode whatever the caller's: evaluate() runs under torch.inference_mode, the pipeline's worker thread
            #  does not -- inference mode is per thread -- and it writes these buffers)
            with torch.inference_mode(False):
                bufs = staging[key] = [torch.empty((N, n_sym), dtype=torch.int32, pin_memory=True),
                                       torch.empty((N, n_sym), dtype=torch.int32, pin_memory=True), None,
                                       torch.empty((2, N), dtype=torch.int32, pin_memory=True)]
        return bufs

    def host_copy_begin(self, sym, staging, slot=0):
        """First half of `code_on_host`, for a caller that wants the device-to-host copy under way before a worker thread picks the
        batch up: enqueues symbols -> pinned host memory on the CURRENT stream, -> the event that marks its end."""
        import time
        t0 = time.perf_counter()
        bufs = self._host_staging(staging, slot, sym.shape[0], sym.shape[1])
        t1 = time.perf_counter()
        if bufs[2] is not None:
            bufs[2].synchronize()          # the previous batch's host-to-device copy out of this slot
        t2 = time.perf_counter()
        bufs[0].copy_(sym, non_blocking=True)      # (copy engine; may block THIS thread for milliseconds while another transfer is in flight)
        t3 = time.perf_counter()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(sym.device))
        if '_trace' in staging:
            staging['_trace'].append({'copy_begin slot': slot, 'staging_ms': 1e3 * (t1 - t0), 'event_sync_ms': 1e3 * (t2 - t1), 'copy_call_ms': 1e3 * (t3 - t2),
                                      'record_ms': 1e3 * (time.perf_counter() - t3)})
        return ev

    def code_on_host(self, sym, size, staging=None, slot=0, d2h_event=None):
        """The coder stage of a whole batch on the HOST thread pool (round 6; pipeline.py `host_steps`): int32 symbols [N, C*hw]
        on the device -> (y_hat bf16 NHWC on the device, nbytes int32 [N], status int32 [N], both on the device): the symbols cross
        to pinned host memory (`d2h_event`: already under way, host_copy_begin), every stream is rANS-encoded into its byte row and
        decoded from it again by the library's host coder (sc2_rans_code_host: one thread per group of streams; the same bytes as
        the batched device coder, tests/test_gpu_host_coder.py), the decoded symbols cross back and are dequantised by the launch
        the device path ends with.  Device work goes to the CURRENT stream; the call blocks its thread while the host codes."""
        dev = sym.device
        N, n_sym = sym.shape
        hw = int(np.prod(size))
        stream = torch.cuda.current_stream(dev)
        staging = staging if staging is not None else {}
        if d2h_event is None:
            d2h_event = self.host_copy_begin(sym, staging, slot)
        bufs = self._host_staging(staging, slot, N, n_sym)
        sym_h, dec_h = bufs[0], bufs[1]
        import time
        t0 = time.perf_counter()
        d2h_event.synchronize()
        t1 = time.perf_counter()
        tables = self._host_tables()
        scratch = staging.setdefault(('scratch', slot), {})       # the byte rows, kept between calls (hip.rans_code_host)
        _, _, nb, st = hip.rans_code_host(tables, sym_h.numpy(), hw, dec_h.numpy(), scratch=scratch)
        if _status_or(st) & 1:     # a row overflowed 2 B / symbol: redo with the proven upper bound
            _, _, nb, st = hip.rans_code_host(tables, sym_h.numpy(), hw, dec_h.numpy(), out_stride=hip.rans_max_bytes(n_sym))
        t2 = time.perf_counter()
        sym_back = torch.empty((N, n_sym), dtype=torch.int32, device=dev)
        sym_back.copy_(dec_h, non_blocking=True)
        # byte counts and status words ride in pinned memory too: a pageable copy would block this thread until the 74 MB in front
        # of it on the stream have crossed
        small = bufs[3]
        small_np = small.numpy()
        small_np[0], small_np[1] = nb, st
        small_d = torch.empty((2, N), dtype=torch.int32, device=dev)
        small_d.copy_(small, non_blocking=True)
        bufs[2] = torch.cuda.Event()
        bufs[2].record(stream)
        y_hat = self.dequantize_device(sym_back, tuple(size))[1]
        if '_trace' in staging:
            staging['_trace'].append({'slot': slot, 'wait_d2h_ms': 1e3 * (t1 - t0), 'host_code_ms': 1e3 * (t2 - t1),
                                      'enqueue_ms': 1e3 * (time.perf_counter() - t2)})
        return y_hat, small_d[0], small_d[1]

    @staticmethod
    def unpack_strings(buf, off, nb):
        """Device streams (end-aligned rows of an encode) -> list[bytes]."""
        nb_h = nb.cpu().numpy()
        off_h = off.cpu().numpy()
        host = buf.cpu().numpy()
        return [host[i, int(off_h[i]):int(off_h[i]) + int(nb_h[i])].tobytes() for i in range(host.shape[0])]

    def compress_symbols(self, sym, hw):
        """int32 symbols [N, C*hw] on the device -> list[bytes], one rANS stream per row.  Up to
        `hip.host_coder_max_streams()` streams (the reference's evaluation mode codes ONE per forward) go through the
        library's HOST coder: a single stream is a serial chain that a CPU core steps ~10x faster than a GPU lane; larger
        batches through the batched device coder.  Same bytes either way (tests/test_gpu_host_coder.py)."""
        N, n_sym = sym.shape
        if 0 < N <= hip.host_coder_max_streams():
            tables = self._host_tables()
            sym_h = sym.cpu().numpy()
            strings, st = hip.rans_encode_host(tables, sym_h, index_div=hw)
            if _status_or(st) & 1:     # a row overflowed 2 B/symbol: redo with the proven upper bound
                strings, st = hip.rans_encode_host(tables, sym_h, index_div=hw, out_stride=hip.rans_max_bytes(n_sym))
            _raise_on_status(torch.from_numpy(st), 'EntropyBottleneck.compress')
            return strings
        cdf, cdf_len, offset = self._tables()
        buf, off, nb, st = hip.rans_encode_batch(sym, cdf, cdf_len, offset, index_div=hw)
        if int(st.max().item()) != 0:
            buf, off, nb, st = hip.rans_encode_batch(sym, cdf, cdf_len, offset, index_div=hw,
                                                     out_stride=hip.rans_max_bytes(n_sym))
            _raise_on_status(st, 'EntropyBottleneck.compress')
        nb_h = nb.cpu().numpy()
        stride = buf.shape[1]
        width = int(nb_h.max())
        tail = buf[:, stride - width:].contiguous().cpu().numpy()  # streams are end-aligned in their rows
        return [tail[i, width - int(nb_h[i]):].tobytes() for i in range(tail.shape[0])]

    def compress(self, x):
        """Returns list[bytes], one rANS stream per batch item (EntropyBottleneck.compress, layer.py:506)."""
        if len(x.size()) < 2:
            raise ValueError('Invalid `inputs` size. Expected a tensor with at least 2 dimensions.')
        _require_device(x, 'EntropyBottleneck.compress')
        cdf = self._tables()[0]
        y = x.float().contiguous()
        N, C = y.shape[0], y.shape[1]
        if C != cdf.shape[0]:
            raise ValueError('`inputs` has {} channels but the CDF table has {} rows'.format(C, cdf.shape[0]))
        if N == 0:
            return []
        hw = y.numel() // (N * C)
        sym = hip.eb_symbols(y, self._median_vector())
        return self.compress_symbols(sym.view(N, C * hw), hw)

    def pack_strings(self, strings, device):
        """list[bytes] -> (buf u8 [N,stride], offset i32 [N], nbytes i32 [N]) on device (start-aligned)."""
        n = len(strings)
        stride = (max(len(s) for s in strings) + 3) // 4 * 4 + 8
        host = np.zeros((n, stride), dtype=np.uint8)
        lens = np.zeros((n,), dtype=np.int32)
        for i, s in enumerate(strings):
            host[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
            lens[i] = len(s)
        buf = torch.from_numpy(host).to(device)
        nb = torch.from_numpy(lens).to(device)
        off = torch.zeros((n,), dtype=torch.int32, device=device)
        return buf, off, nb

    def decompress_device(self, buf, off, nb, size, want_f32=True, want_nhwc=False, check=False):
        """Device buffers -> (y_hat f32 NCHW or None, y_hat bf16 NHWC or None).  `check`: read the decoder's status vector back
        and raise on a corrupt / truncated stream (a host synchronisation: for bytes that arrived from outside, not for streams
        this process has just encoded on the device)."""
        cdf, cdf_len, offset = self._tables()
        C = cdf.shape[0]
        hw = int(np.prod(size))
        N = buf.shape[0]
        sym, st = hip.rans_decode_batch(buf, off, nb, C * hw, cdf, cdf_len, offset, index_div=hw)
        if check:
            _raise_on_status(st, 'EntropyBottleneck.decompress')
        return hip.eb_dequantize(sym.view(N, C, *size), self._median_vector(), want_f32=want_f32,
                                 want_nhwc=want_nhwc)

    def decompress_to_device(self, strings, size, want_f32=True, want_nhwc=False):
        """list[bytes], spatial size -> (y_hat f32 NCHW or None, y_hat bf16 NHWC or None) on the module's device; the host
        coder for a few streams, the batched device coder otherwise (see compress_symbols)."""
        dev = self._quantized_cdf.device
        if dev.type != 'cuda':
            raise hip.Sc2Error('EntropyBottleneck.decompress: module is on {}; HIP device required'.format(dev))
        size = tuple(size)
        if 0 < len(strings) <= hip.host_coder_max_streams():
            C = self._quantized_cdf.shape[0]
            hw = int(np.prod(size))
            sym_h, st_h = hip.rans_decode_host(self._host_tables(), strings, C * hw, index_div=hw)
            _raise_on_status(st_h, 'EntropyBottleneck.decompress')
            sym = torch.from_numpy(sym_h).to(dev)
            return hip.eb_dequantize(sym.view(len(strings), C, *size), self._median_vector(), want_f32=want_f32,
                                     want_nhwc=want_nhwc)
        buf, off, nb = self.pack_strings(strings, dev)
        return self.decompress_device(buf, off, nb, size, want_f32=want_f32, want_nhwc=want_nhwc, check=True)

    def decompress(self, strings, size):
        """list[bytes], spatial size -> f32 [N,C,*size] (EntropyBottleneck.decompress, layer.py:520)."""
        return self.decompress_to_device(strings, size, want_f32=True)[0]


SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """compressai.models.google.get_scale_table (imported by the reference at layer.py:5)."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class GaussianConditional(_HostTablesMixin, nn.Module):
    """Gaussian conditional entropy model with CompressAI 1.2.x semantics, as the hyperprior
    bottlenecks use it (layer.py:627,646-647,665,679,691-693,702,776,785,794,811-813).  forward / quantize /
    dequantize / build_indexes / compress / decompress run in the HIP library (per-symbol CDF rows through the
    explicit-`indexes` path of the batched rANS coder); ``update_scale_table()`` builds the integer tables once
    per model on the host, as the reference does (scipy's normal quantile + the C++ CDF quantiser)."""

    def __init__(self, scale_table=None, *args, scale_bound=0.11, tail_mass=1e-9, likelihood_bound=1e-9,
                 entropy_coder_precision=16, **kwargs):
        super().__init__()
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError('Invalid type for scale_table "{}"'.format(type(scale_table)))
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError('Invalid scale_table length "{}"'.format(len(scale_table)))
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError('Invalid scale_table "({})"'.format(scale_table))
        self.tail_mass = float(tail_mass)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self.likelihood_bound = float(likelihood_bound)
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
        self._scale_bound = float(scale_bound)

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

    _pmf_to_cdf = EntropyBottleneck._pmf_to_cdf
    _check_cdf_size = EntropyBottleneck._check_cdf_size
    _check_offsets_size = EntropyBottleneck._check_offsets_size
    _check_cdf_length = EntropyBottleneck._check_cdf_length
    _tables = EntropyBottleneck._tables
    pack_strings = EntropyBottleneck.pack_strings
    dequantize = staticmethod(EntropyBottleneck.dequantize)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        dev = self.scale_table.device
        table = self.scale_table.detach().cpu()          # host build in f32 torch CPU ops, upstream's op order
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None])
        samples_scale = table.unsqueeze(1)
        samples = samples.float()
        samples_scale = samples_scale.float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._quantized_cdf = quantized_cdf.to(dev)
        self._offset = (-pmf_center).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        self._invalidate_host_tables()

    # ---- quantisation (EntropyModel.quantize; reference calls layer.py:691-693,811-813)
    def quantize(self, inputs, mode, means=None):
        if mode not in ('noise', 'dequantize', 'symbols'):
            raise ValueError('Invalid quantization mode: "{}"'.format(mode))
        _require_device(inputs, 'GaussianConditional.quantize')
        if mode == 'noise':
            half = float(0.5)
            noise = torch.empty_like(inputs).uniform_(-half, half)
            return inputs + noise
        x = inputs.float().contiguous()
        if means is not None:
            means = means.detach().float().expand_as(x)    # raises like upstream's broadcast if the shapes clash
        if mode == 'symbols':
            return hip.gc_symbols_indexes(x, None, means, None, want_indexes=False)[0]
        return hip.gc_forward(x, None, means, mode=hip.EB_DEQUANTIZE, scale_bound=self._scale_bound, want_lik=False)[0]

    def forward(self, inputs, scales, means=None, training=None, noise=None):
        """Returns (outputs, likelihoods), f32 with the input's shape.  ``noise`` overrides the U(-.5,.5) draw."""
        if training is None:
            training = self.training
        _require_device(inputs, 'GaussianConditional.forward')
        x = inputs.float().contiguous()
        bound = self.likelihood_bound if self.use_likelihood_bound else 0.0
        if training and torch.is_grad_enabled() and (x.requires_grad or scales.requires_grad or
                                                      (means is not None and means.requires_grad)):
            from .autograd import gc_forward_autograd
            return gc_forward_autograd(self, x, scales, means, noise)
        if training:
            if noise is None:
                half = float(0.5)
                noise = torch.empty_like(x).uniform_(-half, half)
            return hip.gc_forward(x, scales.float(), None if means is None else means.float(),
                                  noise=noise.float().contiguous(), mode=hip.EB_NOISE, scale_bound=self._scale_bound,
                                  lik_bound=bound)
        return hip.gc_forward(x, scales.float(), None if means is None else means.float(), mode=hip.EB_DEQUANTIZE,
                              scale_bound=self._scale_bound, lik_bound=bound)

    def build_indexes(self, scales):
        _require_device(scales, 'GaussianConditional.build_indexes')
        if self.scale_table.numel() == 0:
            raise ValueError('Uninitialized scale table. Run update() first')
        return hip.gc_symbols_indexes(None, scales.float(), None, self.scale_table.float().contiguous(),
                                      scale_bound=self._scale_bound, want_symbols=False)[1]

    # ---- stage-wise device API (pipeline.StagePipeline: the serial coder on its own HIP stream)
    def symbols_indexes_device(self, inputs, scales, means=None):
        """quantize(inputs, 'symbols', means) and build_indexes(scales) in ONE pass: -> (symbols, indexes), int32 like `inputs`."""
        _require_device(inputs, 'GaussianConditional.symbols_indexes')
        if self.scale_table.numel() == 0:
            raise ValueError('Uninitialized scale table. Run update() first')
        return hip.gc_symbols_indexes(inputs.float().contiguous(), scales.float(), None if means is None else means.float(),
                                      self.scale_table.float().contiguous(), scale_bound=self._scale_bound)

    def encode_symbols_device(self, sym, indexes, out_stride=None):
        """int32 symbols / indexes [N, n] -> (buf, offset, nbytes, status) on the device."""
        cdf, cdf_len, offset = self._tables()
        return hip.rans_encode_batch(sym, cdf, cdf_len, offset, indexes=indexes.int().contiguous(), out_stride=out_stride)

    def decode_symbols_device(self, buf, off, nb, indexes):
        """-> (int32 symbols [N, n], status [N]) on the device."""
        cdf, cdf_len, offset = self._tables()
        return hip.rans_decode_batch(buf, off, nb, indexes.shape[1], cdf, cdf_len, offset, indexes=indexes.int().contiguous())

    # ---- entropy coding on the device
    def compress_device(self, inputs, indexes, means=None, out_stride=None):
        """f32 [N,C,*spatial], int32 indexes -> (buf, offset, nbytes, status) on the device."""
        if len(inputs.size()) < 2:
            raise ValueError('Invalid `inputs` size. Expected a tensor with at least 2 dimensions.')
        if inputs.size() != indexes.size():
            raise ValueError('`inputs` and `indexes` should have the same size.')
        cdf, cdf_len, offset = self._tables()
        x = inputs.float().contiguous()
        sym = hip.gc_symbols_indexes(x, None, None if means is None else means.float(), None, want_indexes=False)[0]
        N = x.shape[0]
        return hip.rans_encode_batch(sym.view(N, -1), cdf, cdf_len, offset,
                                     indexes=indexes.int().contiguous().view(N, -1), out_stride=out_stride)

    def compress(self, inputs, indexes, means=None):
        """-> list[bytes], one rANS stream per batch item (EntropyModel.compress, layer.py:647,776).  A few streams go
        through the library's host coder (EntropyBottleneck.compress_symbols: why)."""
        if 0 < inputs.shape[0] <= hip.host_coder_max_streams():
            if len(inputs.size()) < 2:
                raise ValueError('Invalid `inputs` size. Expected a tensor with at least 2 dimensions.')
            if inputs.size() != indexes.size():
                raise ValueError('`inputs` and `indexes` should have the same size.')
            x = inputs.float().contiguous()
            N = x.shape[0]
            sym = hip.gc_symbols_indexes(x, None, None if means is None else means.float(), None, want_indexes=False)[0]
            sym_h = sym.view(N, -1).cpu().numpy()
            idx_h = indexes.int().contiguous().view(N, -1).cpu().numpy()
            tables = self._host_tables()
            strings, st = hip.rans_encode_host(tables, sym_h, indexes=idx_h)
            if _status_or(st) & 1:
                strings, st = hip.rans_encode_host(tables, sym_h, indexes=idx_h, out_stride=hip.rans_max_bytes(sym_h.shape[1]))
            _raise_on_status(torch.from_numpy(st), 'GaussianConditional.compress')
            return strings
        buf, off, nb, st = self.compress_device(inputs, indexes, means)
        if int(st.max().item()) != 0:
            buf, off, nb, st = self.compress_device(inputs, indexes, means,
                                                    out_stride=hip.rans_max_bytes(inputs[0].numel()))
            _raise_on_status(st, 'GaussianConditional.compress')
        nb_h = nb.cpu().numpy()
        stride = buf.shape[1]
        width = int(nb_h.max())
        tail = buf[:, stride - width:].contiguous().cpu().numpy()
        return [tail[i, width - int(nb_h[i]):].tobytes() for i in range(tail.shape[0])]

    def decompress_device(self, buf, off, nb, indexes, means=None, want_f32=True, want_nhwc=False, check=False):
        """`check`: as EntropyBottleneck.decompress_device (raise on a corrupt / truncated stream; synchronises)."""
        cdf, cdf_len, offset = self._tables()
        N = indexes.shape[0]
        sym, st = hip.rans_decode_batch(buf, off, nb, indexes[0].numel(), cdf, cdf_len, offset,
                                        indexes=indexes.int().contiguous().view(N, -1))
        if check:
            _raise_on_status(st, 'GaussianConditional.decompress')
        return hip.gc_dequantize(sym.view(indexes.shape), None if means is None else means.float(), want_f32=want_f32,
                                 want_nhwc=want_nhwc)

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        """list[bytes], indexes -> f32 tensor shaped like indexes (EntropyModel.decompress, layer.py:665,785)."""
        if not isinstance(strings, (tuple, list)):
            raise ValueError('Invalid `strings` parameter type.')
        if not len(strings) == indexes.size(0):
            raise ValueError('Invalid strings or indexes parameters')
        if means is not None and (means.size()[:2] != indexes.size()[:2]):
            raise ValueError('Invalid means or indexes parameters')
        out = self.decompress_to_device(strings, indexes, means, want_f32=True, want_nhwc=False)[0]
        return out if means is not None else out.type(dtype)

    def decompress_to_device(self, strings, indexes, means=None, want_f32=True, want_nhwc=False):
        """list[bytes], device indexes -> (y_hat f32 NCHW or None, y_hat bf16 NHWC or None); the host coder for a few streams,
        the batched device coder otherwise."""
        dev = indexes.device
        if dev.type != 'cuda':
            raise hip.Sc2Error('GaussianConditional.decompress: indexes are on {}; HIP device required'.format(dev))
        if 0 < len(strings) <= hip.host_coder_max_streams():
            N = indexes.shape[0]
            idx_h = indexes.int().contiguous().view(N, -1).cpu().numpy()
            sym_h, st_h = hip.rans_decode_host(self._host_tables(), list(strings), idx_h.shape[1], indexes=idx_h)
            _raise_on_status(st_h, 'GaussianConditional.decompress')
            sym = torch.from_numpy(sym_h).to(dev)
            return hip.gc_dequantize(sym.view(indexes.shape), None if means is None else means.float(), want_f32=want_f32,
                                     want_nhwc=want_nhwc)
        buf, off, nb = self.pack_strings(strings, dev)
        return self.decompress_device(buf, off, nb, indexes, means, want_f32=want_f32, want_nhwc=want_nhwc, check=True)


class _CpuReplica(object):
    """CPU copies of the bottleneck parameters for the once-per-model table build in update()."""

    def __init__(self, eb):
        self.matrices = [p.detach().cpu().float() for p in eb.matrices]
        self.biases = [p.detach().cpu().float() for p in eb.biases]
        self.factors = [p.detach().cpu().float() for p in eb.factors]

    def logits_cumulative(self, inputs):
        logits = inputs
        n = len(self.matrices)
        for i in range(n):
            logits = torch.matmul(F.softplus(self.matrices[i]), logits)
            logits = logits + self.biases[i]
            if i < n - 1:
                logits = logits + torch.tanh(self.factors[i]) * torch.tanh(logits)
        return logits


# --------------------------------------------------------------------------------------------- #
# CompressionModel
# --------------------------------------------------------------------------------------------- #
def _remap_old_eb_keys(prefix, state_dict):
    """compressai<=1.1 stored `_matrix{i}` / `_bias{i}` / `_factor{i}`; >=1.2 uses ParameterLists."""
    for old, new in (('_matrix', 'matrices'), ('_bias', 'biases'), ('_factor', 'factors')):
        for i in range(5):
            ok = '{}{}{}'.format(prefix, old, i)
            if ok in state_dict:
                state_dict['{}{}.{}'.format(prefix, new, i)] = state_dict.pop(ok)


def update_registered_buffers(module, module_name, buffer_names, state_dict):
    """Resizes registered int buffers to the checkpoint's shapes before loading (compressai.models.utils)."""
    for name in buffer_names:
        key = '{}.{}'.format(module_name, name) if module_name else name
        if key not in state_dict:
            continue
        new = state_dict[key]
        cur = getattr(module, name)
        if cur.shape != new.shape:
            setattr(module, name, torch.empty(new.shape, dtype=cur.dtype, device=cur.device))


class CompressionModel(nn.Module):
    """Base class holding an ``entropy_bottleneck`` (compressai.models.CompressionModel, as used through
    the deprecated ``entropy_bottleneck_channels`` argument at layer.py:409)."""

    def __init__(self, entropy_bottleneck_channels=None, init_weights=None):
        super().__init__()
        if entropy_bottleneck_channels is not None:
            self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)

    def aux_loss(self):
        """Sum of the quantile losses of all EntropyBottleneck children (image_classification.py:76)."""
        loss = sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))
        return loss

    def update(self, scale_table=None, force=False, update_quantiles=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = False
        for m in self.modules():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force, update_quantiles=update_quantiles)
            if isinstance(m, GaussianConditional):
                updated |= m.update_scale_table(scale_table, force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        for name, module in self.named_modules():
            if isinstance(module, EntropyBottleneck):
                prefix = name + '.' if name else ''
                _remap_old_eb_keys(prefix, state_dict)
                update_registered_buffers(module, name, ['_quantized_cdf', '_offset', '_cdf_length'], state_dict)
        return nn.Module.load_state_dict(self, state_dict, strict=strict)
