"""Training path: autograd wiring of the HIP forward kernels.

FORWARD runs on the hand-written kernels (same launches as inference).  BACKWARD in this round is INTERIM:
gradients are computed on the device with PyTorch-ROCm ops (MIOpen `convolution_backward` for the convs,
re-evaluation of the GDN / entropy-bottleneck formulas under torch autograd for the rest).  Hand-written
dgrad / wgrad / GDN-bwd / bottleneck-bwd kernels are the next row of DESIGN.md section 8; nothing here touches the
CPU or the oracle.

Reference semantics reproduced: `_forward2train` (sc2bench/models/layer.py:529-533) before `update()`, and the
round + detach path after it (layer.py:543-549); `LowerBound` gradient rule of CompressAI (gradient passes where
x >= bound or where it pushes x up).
"""
import torch
import torch.nn.functional as F

from . import hip


def _cl(t_nhwc):
    """bf16 [N,H,W,C] -> logical [N,C,H,W] view with channels_last strides (no copy)."""
    return t_nhwc.permute(0, 3, 1, 2)


class _ToNhwcBf16(torch.autograd.Function):
    """f32 NCHW -> bf16 NHWC (channels padded to `cpad`)."""

    @staticmethod
    def forward(ctx, x, cpad):
        ctx.c = x.shape[1]
        return hip.nchw_f32_to_nhwc_bf16(x.float().contiguous(), cpad)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.shape[-1] % 8 == 0:
            gx = hip.nhwc_bf16_to_nchw_f32(g)
        else:
            gx = g.float().permute(0, 3, 1, 2).contiguous()
        return gx[:, :ctx.c].contiguous(), None


class _ConvFn(torch.autograd.Function):
    """y = conv(x) on the implicit-GEMM kernel; x bf16 NHWC, weight the f32 OIHW parameter."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, packed, kh, kw, stride, pad, out_format, tag, w_view):
        y = hip.conv2d_fwd(x_nhwc, packed, weight.shape[0], kh, kw, stride, pad, out_format=out_format, tag=tag)
        ctx.save_for_backward(x_nhwc, weight)
        ctx.cfg = (stride, pad, out_format, w_view)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight = ctx.saved_tensors
        stride, pad, out_format, w_view = ctx.cfg
        if out_format == hip.OUT_BF16_NHWC:
            g = _cl(gy.contiguous())
        elif out_format == hip.OUT_F32_NCHW:
            g = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        else:
            g = _cl(gy.to(torch.bfloat16).contiguous())
        # the kernel may see the input through a different (Cin, KW) view than the parameter (first encoder conv)
        x_log, w_log = w_view(x_nhwc, weight) if w_view is not None else (_cl(x_nhwc), weight)
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        ph, pw = (pad, pad) if isinstance(pad, int) else pad
        if w_view is not None:
            sh, sw, ph, pw = w_view.stride + w_view.pad
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gi, gw, _ = torch.ops.aten.convolution_backward(
            g, x_log, w_log.to(torch.bfloat16), None, (sh, sw), (ph, pw), (1, 1), False, (0, 0), 1,
            (need_x, need_w, False))
        if need_w:
            gw = gw.float()
            if w_view is not None:
                gw = gw[:, :weight.shape[1]]
        if need_x:
            gi = gi.permute(0, 2, 3, 1).contiguous()
        return gi if need_x else None, gw if need_w else None, None, None, None, None, None, None, None, None


class _PairView(object):
    """Backward-side view of the pixel-pair first conv: the stored input is NHWC with 4 channels; the parameter is
    [Cout, 3, 5, 5] with stride 2 / pad 2."""
    stride = (2, 2)
    pad = (2, 2)

    def __call__(self, x_pairs, weight):
        n, h, w2, _ = x_pairs.shape
        x4 = x_pairs.view(n, h, w2 * 2, 4)
        w4 = torch.zeros((weight.shape[0], 4) + tuple(weight.shape[2:]), dtype=weight.dtype, device=weight.device)
        w4[:, :weight.shape[1]] = weight
        return _cl(x4), w4


class _GdnFn(torch.autograd.Function):
    """GDN1 / inverse GDN1 on the fused 1x1-GEMM kernel.  beta [C] f32 and gamma [C,C] f32 are the EFFECTIVE
    (reparametrised) tensors, so the reparametrisation itself stays in torch autograd."""

    @staticmethod
    def forward(ctx, x_nhwc, beta, gamma, inverse, tag):
        C = beta.numel()
        packed = hip.pack_conv_weight(gamma.reshape(C, C, 1, 1))
        y = hip.conv2d_fwd(x_nhwc, packed, C, 1, 1, 1, 0, a_op=hip.AOP_ABS,
                           epilogue=hip.EPI_IGDN if inverse else hip.EPI_GDN, ep_x=x_nhwc,
                           ep_beta=beta.float().contiguous(), tag=tag)
        ctx.save_for_backward(x_nhwc, beta, gamma)
        ctx.inverse = inverse
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, beta, gamma = ctx.saved_tensors
        C = beta.numel()
        with torch.enable_grad():
            x = _cl(x_nhwc).detach().requires_grad_(True)
            b = beta.detach().to(torch.bfloat16).requires_grad_(True)
            g = gamma.detach().to(torch.bfloat16).requires_grad_(True)
            norm = F.conv2d(torch.abs(x), g.reshape(C, C, 1, 1), b)
            y = x * norm if ctx.inverse else x / norm
            gx, gb, gg = torch.autograd.grad(y, (x, b, g), _cl(gy.contiguous()))
        return gx.permute(0, 2, 3, 1).contiguous(), gb.float(), gg.float(), None, None


def _eb_logits(v, P):
    """Cumulative logits from the packed effective-parameter block P [C,64]; v: [C,1,T]."""
    C = P.shape[0]
    h = P[:, 0:3].reshape(C, 3, 1) * v + P[:, 3:6].reshape(C, 3, 1)
    h = h + P[:, 6:9].reshape(C, 3, 1) * torch.tanh(h)
    for layer in range(3):
        o = 9 + 15 * layer
        h = torch.matmul(P[:, o:o + 9].reshape(C, 3, 3), h) + P[:, o + 9:o + 12].reshape(C, 3, 1)
        h = h + P[:, o + 12:o + 15].reshape(C, 3, 1) * torch.tanh(h)
    return torch.matmul(P[:, 54:57].reshape(C, 1, 3), h) + P[:, 57:58].reshape(C, 1, 1)


class _LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = bound
        return torch.clamp(x, min=bound)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ((x >= ctx.bound) | (g < 0)) * g, None


class _EbFn(torch.autograd.Function):
    """(y_hat, likelihood) of the factorised-prior bottleneck on the fused element-wise kernel."""

    @staticmethod
    def forward(ctx, y, params, noise, training, lik_bound):
        mode = hip.EB_NOISE if training else hip.EB_DEQUANTIZE
        y_hat, _, lik, _ = hip.eb_forward(y, params.detach().contiguous(), mode, noise=noise, lik_bound=lik_bound)
        ctx.save_for_backward(y, params, noise if noise is not None else y.new_zeros(1))
        ctx.training = training
        ctx.lik_bound = lik_bound
        return y_hat, lik

    @staticmethod
    def backward(ctx, g_yhat, g_lik):
        y, params, noise = ctx.saved_tensors
        mode = hip.EB_NOISE if ctx.training else hip.EB_DEQUANTIZE
        gy, gp = hip.eb_backward(y, params.detach().contiguous(), mode, noise if ctx.training else None,
                                 g_yhat.float().contiguous() if g_yhat is not None else None,
                                 g_lik.float().contiguous() if g_lik is not None else None, lik_bound=ctx.lik_bound)
        return gy, gp, None, None, None


def eb_forward_autograd(eb, y, training, noise=None):
    """EntropyBottleneck.forward with gradients (called from entropy.py when grad is enabled)."""
    params = eb.effective_params()        # differentiable torch ops on the 58 parameters per channel
    if training and noise is None:
        half = float(0.5)
        noise = torch.empty_like(y).uniform_(-half, half)
    bound = eb.likelihood_bound if eb.use_likelihood_bound else 0.0
    return _EbFn.apply(y, params, noise.float().contiguous() if noise is not None else None, bool(training), bound)


def _conv(mod, x_nhwc, out_format=hip.OUT_BF16_NHWC):
    return _ConvFn.apply(x_nhwc, mod.weight, mod.packed_weight(), mod.kernel_size[0], mod.kernel_size[1],
                         mod.stride, mod.padding, out_format, getattr(mod, '_tag', None), None)


def _gdn(mod, x_nhwc):
    beta = mod.beta_reparam(mod.beta)
    gamma = mod.gamma_reparam(mod.gamma)
    return _GdnFn.apply(x_nhwc, beta, gamma, mod.inverse, getattr(mod, '_tag', None))


def analysis_autograd(m, x):
    """encoder(x) with gradients: f32 NCHW image -> f32 NCHW latent."""
    c0, g1, c2, g3, c4 = m.encoder
    x = x.float()
    if m._uses_pair_conv0(x):
        N, _, H, W = x.shape
        x4 = _ToNhwcBf16.apply(x, 4)
        xp = x4.view(N, H, W // 2, 8)
        h = _ConvFn.apply(xp, c0.weight, m._conv0_packed(), 5, 3, (2, 1), (2, 1), hip.OUT_BF16_NHWC, c0._tag,
                          _PairView())
    else:
        cin = c0.in_channels
        if cin % 8 != 0:
            raise hip.Sc2Error('training path: first conv needs the pixel-pair form (Cin<=4, k5 s2 p2, even width) '
                               'or Cin % 8 == 0')
        h = _conv(c0, _ToNhwcBf16.apply(x, cin))
    h = _gdn(g1, h)
    h = _conv(c2, h)
    h = _gdn(g3, h)
    return _conv(c4, h, hip.OUT_F32_NCHW)


def synthesis_autograd(m, y_hat):
    """decoder(y_hat) with gradients: f32 NCHW latent -> f32 NCHW features."""
    c0, g1, c2, g3, c4 = m.decoder
    h = _ToNhwcBf16.apply(y_hat, y_hat.shape[1])
    h = _conv(c0, h)
    h = _gdn(g1, h)
    h = _conv(c2, h)
    h = _gdn(g3, h)
    return _conv(c4, h, hip.OUT_F32_NCHW)


def bottleneck_forward2train_autograd(m, x):
    """layer.py:529-533: encoder -> entropy_bottleneck (its forward hook sees (y_hat, likelihoods)) -> decoder."""
    y = analysis_autograd(m, x)
    y_hat, _ = m.entropy_bottleneck(y)
    return synthesis_autograd(m, y_hat)


def bottleneck_forward_updated_autograd(m, x):
    """layer.py:543-549: after update(), training decodes round(y - median) + median, detached."""
    with torch.no_grad():
        y = m.analysis(x)
        y_hat = m.entropy_bottleneck.quantize(y, 'dequantize', m._get_means(y))
    return synthesis_autograd(m, y_hat.detach())
