"""Training path: autograd wiring of the HIP forward kernels.

FORWARD and BACKWARD both run on the hand-written kernels: data gradients of the convs on the forward
implicit-GEMM kernel (flipped sub-filters per stride-parity class, `hip.conv2d_dgrad`), weight gradients on
`conv_wgrad.hip` (transposing LDS reads), GDN1 backward as two element-wise kernels around a gamma^T GEMM and a
wgrad, the entropy bottleneck on `eb_backward_kernel`.  torch autograd only stitches the pieces together and
differentiates the parameter-sized reparametrisations (beta/gamma lower bounds, softplus/tanh of the 58
bottleneck parameters per channel).  The frozen ResNet head's input gradient still goes through torch modules.

Reference semantics reproduced: `_forward2train` (sc2bench/models/layer.py:529-533) before `update()`, and the
round + detach path after it (layer.py:543-549); `LowerBound` gradient rule of CompressAI (gradient passes where
x >= bound or where it pushes x up).
"""
import torch

from . import hip


def _backward_pass_id():
    """id of the autograd graph task the caller's backward() runs in (-1 outside a backward pass)."""
    return torch._C._current_graph_task_id()


class MseSink(list):
    """Hand-over list between the MSE nodes on a producer's output and the producer's own backward.  Every entry is tagged with
    the backward pass (autograd graph task) that wrote it: the producer applies the entries of ITS pass only and drops the rest,
    so an entry left behind by a pass in which the producer's backward never ran (`torch.autograd.grad(loss, inputs=[feature])`,
    an exception mid-backward) is neither applied later nor applied twice on a retained graph.

    Restriction (documented in DESIGN.md section 7): with the sink in use the tensor-level gradient of the producer's output does
    not carry the MSE term -- `mse_fast_path` therefore hands no sink to an output that retains its gradient or carries tensor
    hooks AT THE TIME THE LOSS IS BUILT; register such hooks before computing the loss, or switch `host_policy.mse_fused` off."""

    def put(self, entry):
        self.append((_backward_pass_id(), entry))

    def drain(self):
        now = _backward_pass_id()
        mine = [e for pid, e in self if pid == now]
        self.clear()
        return mine


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


def _pair_conv0_wgrad(x_pairs, g, weight):
    """Weight gradient of the first encoder conv (Cin <= 4, k5 s2 p2) from its pixel-pair input view [N, H, W/2, 8]: there the conv
    is k (5, 3), stride (2, 1), pad (2, 1) with K ordered (kh, pair tap, pixel-in-pair * 4 + channel)."""
    raw = hip.conv2d_wgrad(x_pairs, g, 5, 3, (2, 1), (2, 1))            # [Cout, 8, 5, 3]
    cout = raw.shape[0]
    full = raw.permute(0, 2, 3, 1).reshape(cout, 5, 3, 2, 4)            # (kh, t, dw, c)
    return full.reshape(cout, 5, 6, 4)[:, :, :5, :weight.shape[1]].permute(0, 3, 1, 2).contiguous()


class _ConvFn(torch.autograd.Function):
    """y = conv(x) on the implicit-GEMM kernel; x bf16 NHWC, weight the f32 OIHW parameter."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, packed, kh, kw, stride, pad, out_format, tag, w_view, k_order=0, win_pad=None):
        if win_pad is not None:      # `packed` is the window-plane kernel's weight stream (bit-identical to the tile kernel)
            y = hip.conv2x2_win_fwd(x_nhwc, packed, win_pad, tag=tag)
        else:
            y = hip.conv2d_fwd(x_nhwc, packed, weight.shape[0], kh, kw, stride, pad, out_format=out_format, tag=tag,
                               k_order=k_order)
        ctx.save_for_backward(x_nhwc, weight)
        ctx.cfg = (stride, pad, out_format, w_view)
        # a feature-matching MSE term on this output may hand (its operands, scale) over instead of a gradient tensor
        # (frozen.MseSumFn.backward): the sink travels on the output (and on the channels_last view synthesis_autograd returns)
        ctx.mse_sink = MseSink()
        if out_format == hip.OUT_BF16_NHWC:
            y._sc2_mse_sink = ctx.mse_sink
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight = ctx.saved_tensors
        stride, pad, out_format, w_view = ctx.cfg
        sink = ctx.mse_sink.drain()      # (the MSE nodes keep the list itself: emptied, not replaced -- a retained graph may run
                                         #  again; entries of another backward pass are dropped, frozen.MseSink)
        if out_format == hip.OUT_BF16_NHWC:
            g = gy.contiguous()
            for xs, ts, scale in sink:           # 2 scale (y - t) added in one pass (no gradient tensor, no separate add)
                ys, tt = xs.permute(0, 2, 3, 1), ts.permute(0, 2, 3, 1)
                if hip._same_dense_bf16(g, ys, tt):
                    g = hip.relu_bwd_mse(g, ys, tt, scale, relu=False)
                else:
                    g = g + hip.mse_grad(xs, ts, scale).permute(0, 2, 3, 1)
        elif out_format == hip.OUT_F32_NCHW:
            g = hip.nchw_f32_to_nhwc_bf16(gy.float().contiguous())
        else:
            g = gy.to(torch.bfloat16).contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gi = gw = None
        if w_view is not None:
            if need_w:
                gw = _pair_conv0_wgrad(x_nhwc, g, weight)
            if need_x:
                raise hip.Sc2Error('the pixel-pair first conv has no data gradient (its input is the image)')
        else:
            kh, kw = weight.shape[2], weight.shape[3]
            if need_w:
                gw = hip.conv2d_wgrad(x_nhwc, g, kh, kw, stride, pad).contiguous()
            if need_x:
                gi = hip.conv2d_dgrad(g, weight, stride, pad, (x_nhwc.shape[1], x_nhwc.shape[2]))
        return gi, gw, None, None, None, None, None, None, None, None, None, None


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
        ctx.save_for_backward(x_nhwc, beta, gamma)
        ctx.inverse = inverse
        if hip.gdn1_rows_supported(x_nhwc, C):      # C = 256 / 512: the resident-row kernel (gdn512_rows.hip)
            return hip.gdn1_rows_fwd(x_nhwc, hip.pack_weight_fragments(gamma.detach()), beta.detach().float().contiguous(), inverse, tag=tag)
        packed = hip.pack_conv_weight(gamma.reshape(C, C, 1, 1))
        return hip.conv2d_fwd(x_nhwc, packed, C, 1, 1, 1, 0, a_op=hip.AOP_ABS,
                              epilogue=hip.EPI_IGDN if inverse else hip.EPI_GDN, ep_x=x_nhwc,
                              ep_beta=beta.float().contiguous(), tag=tag)

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, beta, gamma = ctx.saved_tensors
        dx, d_beta, d_gamma = hip.gdn1_backward(gy, x_nhwc, beta, gamma, ctx.inverse)
        return dx, d_beta, d_gamma, None, None


class _Conv2Gdn48Fn(torch.autograd.Function):
    """encoder[2] + GDN1(48) of the training forward as the fused inference launch (conv2_gdn48.hip, 0.27 ms at bs 256 against
    0.49 + 0.06 for the two unfused kernels), which for this caller also writes the conv output t in front of the GDN: the one
    tensor the GDN's backward needs and the fused launch otherwise never materialises.  The backward is the unfused one, piece by
    piece: GDN backward on (gy, t), then the conv's weight and data gradients on its result.  (y is computed from the f32 conv
    output, the saved t is its bf16 rounding -- the value the unfused forward would have normalised: the gradient is taken at a
    point half a bf16 ulp from the forward's, which is inside the tolerance every bf16 gradient of this path carries.)"""

    @staticmethod
    def forward(ctx, x_nhwc, weight, beta, gamma, conv, owner, inverse, tag):
        with torch.no_grad():
            gfrag = owner._gdn48_fragments(hip.pack_conv_weight(gamma.detach().reshape(48, 48, 1, 1)))
        y, t = hip.conv2_gdn48_fwd(x_nhwc, conv.packed_weight(hip.K_SLAB_MAJOR | hip.K_B_FRAG_MAJOR), gfrag,
                                   beta.detach().float().contiguous(), inverse, tag=tag, want_t=True)
        ctx.save_for_backward(x_nhwc, weight, t, beta, gamma)
        ctx.cfg = (conv.stride, conv.padding, inverse)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight, t, beta, gamma = ctx.saved_tensors
        stride, pad, inverse = ctx.cfg
        g, d_beta, d_gamma = hip.gdn1_backward(gy.contiguous(), t, beta, gamma, inverse)
        gw = gi = None
        if ctx.needs_input_grad[1]:
            gw = hip.conv2d_wgrad(x_nhwc, g, weight.shape[2], weight.shape[3], stride, pad).contiguous()
        if ctx.needs_input_grad[0]:
            gi = hip.conv2d_dgrad(g, weight, stride, pad, (x_nhwc.shape[1], x_nhwc.shape[2]))
        return gi, gw, d_beta, d_gamma, None, None, None, None


class _Conv0Gdn96Fn(torch.autograd.Function):
    """encoder[0] + GDN1(96) of the training forward as the fused inference launch on the pixel-pair view (conv0_gdn96.hip), which for
    this caller also writes the conv output t (0.25 + 0.25 ms for the two unfused kernels at bs 256).  Backward: the GDN's on (gy, t)
    -- the strips kernel -- and the conv's weight gradient from the pair view (its input is the image: no data gradient)."""

    @staticmethod
    def forward(ctx, x_pairs, weight, beta, gamma, owner, inverse, tag):
        y, t = hip.conv0_gdn96_fwd(x_pairs, owner._conv0_fragments(), hip.pack_gamma_fragments(gamma.detach()),
                                   beta.detach().float().contiguous(), inverse, tag=tag, want_t=True)
        ctx.save_for_backward(x_pairs, weight, t, beta, gamma)
        ctx.inverse = inverse
        return y

    @staticmethod
    def backward(ctx, gy):
        x_pairs, weight, t, beta, gamma = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise hip.Sc2Error('the pixel-pair first conv has no data gradient (its input is the image)')
        g, d_beta, d_gamma = hip.gdn1_backward(gy.contiguous(), t, beta, gamma, ctx.inverse)
        gw = _pair_conv0_wgrad(x_pairs, g, weight) if ctx.needs_input_grad[1] else None
        return None, gw, d_beta, d_gamma, None, None, None


class _Dec0Gdn512Fn(torch.autograd.Function):
    """decoder[0] + (inverse) GDN1(512) of the training forward as the fused inference launch (conv_gdn512.hip), which for this
    caller also writes the conv output t (0.26 + 0.59 ms for the two unfused kernels at bs 256).  Backward as in _Conv2Gdn48Fn:
    the GDN's backward on (gy, t) -- the resident-row kernel -- and the conv's weight / data gradients on its result."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, beta, gamma, conv, inverse, tag):
        y, t = hip.conv2x2_gdn512_fwd(x_nhwc, conv.packed_weight(hip.K_TAP_MAJOR), hip.pack_gamma_fragments(gamma.detach()),
                                      beta.detach().float().contiguous(), inverse, tag=tag, want_t=True)
        ctx.save_for_backward(x_nhwc, weight, t, beta, gamma)
        ctx.cfg = (conv.stride, conv.padding, inverse)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight, t, beta, gamma = ctx.saved_tensors
        stride, pad, inverse = ctx.cfg
        g, d_beta, d_gamma = hip.gdn1_backward(gy.contiguous(), t, beta, gamma, inverse)
        gw = gi = None
        if ctx.needs_input_grad[1]:
            gw = hip.conv2d_wgrad(x_nhwc, g, weight.shape[2], weight.shape[3], stride, pad).contiguous()
        if ctx.needs_input_grad[0]:
            gi = hip.conv2d_dgrad(g, weight, stride, pad, (x_nhwc.shape[1], x_nhwc.shape[2]))
        return gi, gw, d_beta, d_gamma, None, None, None


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
    kh, kw = mod.kernel_size
    # round 5: the decoder's last conv (256 -> 256, k2, p1) of the TRAINING forward on the window-plane kernel the inference path
    # uses (1.13 -> 0.36 ms at bs 256; same products in the same order as the tile kernel: tests/test_gpu_kernels.py::
    # test_conv2x2_win); the 512 -> 256 conv stays on the persistent tile kernel, which is faster without the fused GDN
    if (out_format == hip.OUT_BF16_NHWC and mod.bias is None and mod.in_channels == 256 and mod.out_channels == 256 and
            hip.conv2x2_win_supported(tuple(x_nhwc.shape), 256, kh, kw, mod.stride, mod.padding)):
        with torch.no_grad():
            stream = hip.pack_conv2x2_win(mod.weight)
        return _ConvFn.apply(x_nhwc, mod.weight, stream, kh, kw, mod.stride, mod.padding, out_format, getattr(mod, '_tag', None),
                             None, 0, int(mod.padding[0]))
    return _ConvFn.apply(x_nhwc, mod.weight, mod.packed_weight(), kh, kw,
                         mod.stride, mod.padding, out_format, getattr(mod, '_tag', None), None, mod.k_order())


def _gdn(mod, x_nhwc):
    beta = mod.beta_reparam(mod.beta)
    gamma = mod.gamma_reparam(mod.gamma)
    return _GdnFn.apply(x_nhwc, beta, gamma, mod.inverse, getattr(mod, '_tag', None))


def analysis_autograd(m, x):
    """encoder(x) with gradients: f32 NCHW image -> f32 NCHW latent."""
    c0, g1, c2, g3, c4 = m._g_a()
    x = x.float()
    if m._uses_pair_conv0(x):
        if x.shape[-1] % 2:     # odd width: one zero column (what the conv's own padding reads)
            x = torch.nn.functional.pad(x, (0, 1))
        N, _, H, W = x.shape
        x4 = _ToNhwcBf16.apply(x, 4)
        xp = x4.view(N, H, W // 2, 8)
        fused0 = (hip.host_policy.train_fused_conv0 and g1.in_channels == 96 and c0.out_channels == 96 and
                  hip.conv0_gdn96_supported(tuple(xp.shape), c0.out_channels))
        if fused0:
            h = _Conv0Gdn96Fn.apply(xp, c0.weight, g1.beta_reparam(g1.beta), g1.gamma_reparam(g1.gamma), m, g1.inverse,
                                    c0._tag + '+' + g1._tag)
        else:
            h = _ConvFn.apply(xp, c0.weight, m._conv0_packed(), 5, 3, (2, 1), (2, 1), hip.OUT_BF16_NHWC, c0._tag,
                              _PairView(), hip.K_TAP_MAJOR)
    else:
        fused0 = False
        cin = c0.in_channels
        if cin % 8 != 0:
            raise hip.Sc2Error('training path: first conv needs the pixel-pair form (Cin<=4, k5 s2 p2, even width) '
                               'or Cin % 8 == 0')
        h = _conv(c0, _ToNhwcBf16.apply(x, cin))
    if not fused0:
        h = _gdn(g1, h)
    if (hip.host_policy.train_fused_conv2 and g3.in_channels == 48 and c2.bias is None and
            hip.conv2_gdn48_supported(tuple(h.shape), c2.out_channels, c2.kernel_size[0], c2.kernel_size[1], c2.stride, c2.padding)):
        h = _Conv2Gdn48Fn.apply(h, c2.weight, g3.beta_reparam(g3.beta), g3.gamma_reparam(g3.gamma), c2, m, g3.inverse,
                                c2._tag + '+' + g3._tag)
    else:
        h = _conv(c2, h)
        h = _gdn(g3, h)
    return _conv(c4, h, hip.OUT_F32_NCHW)


def synthesis_autograd(m, y_hat):
    """decoder(y_hat) with gradients: f32 NCHW latent -> f32 NCHW features (or a bf16 channels_last view, per output_format)."""
    c0, g1, c2, g3, c4 = m._g_s()
    h = _ToNhwcBf16.apply(y_hat, y_hat.shape[1])
    n_px = h.shape[0] * (h.shape[1] + 1) * (h.shape[2] + 1)
    if (hip.host_policy.train_fused_dec0 and g1.in_channels == c0.out_channels and c0.bias is None and n_px * 1024 < 2 ** 31 and
            hip.conv2x2_gdn512_supported(c0.in_channels, c0.out_channels, c0.kernel_size[0], c0.kernel_size[1], c0.stride, c0.padding)):
        h = _Dec0Gdn512Fn.apply(h, c0.weight, g1.beta_reparam(g1.beta), g1.gamma_reparam(g1.gamma), c0, g1.inverse,
                                c0._tag + '+' + g1._tag)
    else:
        h = _conv(c0, h)
        h = _gdn(g1, h)
    h = _conv(c2, h)
    h = _gdn(g3, h)
    if getattr(m, 'output_format', 'f32_nchw') == 'bf16_nhwc':     # a bf16 channels_last view for a bf16 tail / loss
        y = _conv(c4, h, hip.OUT_BF16_NHWC)
        res = _cl(y)
        sink = getattr(y, '_sc2_mse_sink', None)
        if sink is not None:
            res._sc2_mse_sink = sink           # (entries for THIS producer carry the MSE's own operands: MseSumFn appends (x, y, scale))
            res._sc2_mse_sink_wants_x = True
        return res
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


# --------------------------------------------------------------------------------------------- #
# hyperprior bottlenecks (layer.py:553-817) under autograd
# --------------------------------------------------------------------------------------------- #
_ACT_SLOPE = {0: None, 1: 0.0, 2: 0.01}     # none / ReLU / LeakyReLU(0.01)


def _act_backward(g, out, act):
    """Gradient through the activation fused into a conv epilogue; ReLU and LeakyReLU keep the sign, so the mask
    comes from the saved OUTPUT."""
    if act == 0:
        return g
    slope = _ACT_SLOPE[act]
    return torch.where(out > 0, g, g * slope) if slope else torch.where(out > 0, g, torch.zeros_like(g))


class _ConvActFn(torch.autograd.Function):
    """conv (+ fused ReLU / LeakyReLU) of h_a / h_s on the implicit-GEMM kernel; x bf16 NHWC, bf16 NHWC or f32 NCHW out."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, mod, act, out_format):
        epi = {0: hip.EPI_NONE, 1: hip.EPI_BIAS_RELU, 2: hip.EPI_BIAS_LEAKY_RELU}[act]
        beta = torch.zeros(mod.out_channels, dtype=torch.float32, device=x_nhwc.device) if act else None
        y = hip.conv2d_fwd(x_nhwc, mod.packed_weight(), mod.out_channels, mod.kernel_size[0], mod.kernel_size[1],
                           mod.stride, mod.padding, epilogue=epi, ep_beta=beta, out_format=out_format,
                           tag=getattr(mod, '_tag', None), k_order=mod.k_order())
        ctx.save_for_backward(x_nhwc, weight, y if act else x_nhwc.new_zeros(1))
        ctx.cfg = (mod.stride, mod.padding, act, out_format)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight, y = ctx.saved_tensors
        stride, pad, act, out_format = ctx.cfg
        gy = _act_backward(gy, y, act)
        if out_format == hip.OUT_BF16_NHWC:
            g = gy.contiguous()
        elif out_format == hip.OUT_F32_NCHW:
            g = hip.nchw_f32_to_nhwc_bf16(gy.float().contiguous())
        else:
            g = gy.to(torch.bfloat16).contiguous()
        kh, kw = weight.shape[2], weight.shape[3]
        gw = hip.conv2d_wgrad(x_nhwc, g, kh, kw, stride, pad).contiguous() if ctx.needs_input_grad[1] else None
        gi = hip.conv2d_dgrad(g, weight, stride, pad, (x_nhwc.shape[1], x_nhwc.shape[2])) \
            if ctx.needs_input_grad[0] else None
        return gi, gw, None, None, None


class _ConvTransposeActFn(torch.autograd.Function):
    """ConvTranspose2d (+ fused activation) of h_s: forward as stride-parity classes with output scatter; backward:
    the data gradient is the plain strided convolution with the same weights, the weight gradient the conv weight
    gradient with the roles of input and output gradient exchanged."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, mod, act):
        epi = {0: hip.EPI_NONE, 1: hip.EPI_BIAS_RELU, 2: hip.EPI_BIAS_LEAKY_RELU}[act]
        beta = torch.zeros(mod.out_channels, dtype=torch.float32, device=x_nhwc.device) if act else None
        y = mod.forward_nhwc(x_nhwc, epi, beta)
        ctx.save_for_backward(x_nhwc, weight, y if act else x_nhwc.new_zeros(1))
        ctx.cfg = (mod.stride, mod.padding, act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight, y = ctx.saved_tensors
        stride, pad, act = ctx.cfg
        g = _act_backward(gy, y, act).contiguous()
        cin, cout, kh, kw = weight.shape                      # [in, out, kh, kw] = a conv weight [Cout'=in, Cin'=out]
        gi = gw = None
        if ctx.needs_input_grad[0]:
            gi = hip.conv2d_fwd(g, hip.pack_conv_weight(weight.detach()), cin, kh, kw, stride, pad, tag='convT.dgrad')
        if ctx.needs_input_grad[1]:
            gw = hip.conv2d_wgrad(g, x_nhwc, kh, kw, stride, pad).contiguous()
        return gi, gw, None, None


def hyper_sequence_autograd(seq, x_nhwc):
    """h_a / h_s (HipConv2d / HipConvTranspose2d / ReLU / LeakyReLU) with gradients: bf16 NHWC in, f32 NCHW out."""
    from .entropy import HipConv2d, HipConvTranspose2d
    mods = list(seq)
    h = x_nhwc
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        act = 0
        if isinstance(nxt, torch.nn.ReLU):
            act = 1
        elif isinstance(nxt, torch.nn.LeakyReLU) and abs(nxt.negative_slope - 0.01) < 1e-12:
            act = 2
        last = i + (2 if act else 1) >= len(mods)
        if isinstance(m, HipConv2d):
            h = _ConvActFn.apply(h, m.weight, m, act, hip.OUT_F32_NCHW if last else hip.OUT_BF16_NHWC)
        elif isinstance(m, HipConvTranspose2d):
            h = _ConvTransposeActFn.apply(h, m.weight, m, act)
            if last:
                h = h.float().permute(0, 3, 1, 2).contiguous()
        else:
            raise hip.Sc2Error('training path of the hyperprior bottlenecks: unsupported module {} in h_a / h_s'
                               .format(type(m).__name__))
        i += 2 if act else 1
    return h


class _GcFn(torch.autograd.Function):
    """(y_hat, likelihood) of the Gaussian conditional model in training (noise) mode on the fused kernels."""

    @staticmethod
    def forward(ctx, y, scales, means, noise, scale_bound, lik_bound):
        y_hat, lik = hip.gc_forward(y, scales, means, noise=noise, mode=hip.EB_NOISE, scale_bound=scale_bound,
                                    lik_bound=lik_bound)
        ctx.save_for_backward(y, scales, means if means is not None else y.new_zeros(1), noise)
        ctx.has_means = means is not None
        ctx.bounds = (scale_bound, lik_bound)
        return y_hat, lik

    @staticmethod
    def backward(ctx, g_yhat, g_lik):
        y, scales, means, noise = ctx.saved_tensors
        g_y, g_s, g_m = hip.gc_backward(y, scales, means if ctx.has_means else None, noise,
                                        g_yhat.float().contiguous() if g_yhat is not None else None,
                                        g_lik.float().contiguous() if g_lik is not None else None,
                                        scale_bound=ctx.bounds[0], lik_bound=ctx.bounds[1])
        return g_y, g_s, g_m, None, None, None


def gc_forward_autograd(gc, y, scales, means, noise=None):
    """GaussianConditional.forward with gradients, training (noise) mode."""
    y = y.float().contiguous()
    if noise is None:
        half = float(0.5)
        noise = torch.empty_like(y).uniform_(-half, half)
    bound = gc.likelihood_bound if gc.use_likelihood_bound else 0.0
    return _GcFn.apply(y, scales.float().contiguous(), None if means is None else means.float().contiguous(),
                       noise.float().contiguous(), gc._scale_bound, bound)


def hyperprior_forward2train_autograd(m, x, noise_z=None, noise_y=None):
    """layer.py:673-680 / 788-795: g_a -> h_a -> entropy_bottleneck -> h_s -> gaussian_conditional -> g_s; both entropy
    modules are called as modules so that their forward hooks see (outputs, likelihoods) for the rate terms."""
    y = analysis_autograd(m, x)
    hin = torch.abs(y) if m._hyper_abs else y
    z = hyper_sequence_autograd(m.h_a, _ToNhwcBf16.apply(hin, hin.shape[1]))
    z_hat, z_lik = m.entropy_bottleneck(z, noise=noise_z)
    params = hyper_sequence_autograd(m.h_s, _ToNhwcBf16.apply(z_hat, z_hat.shape[1]))
    scales_hat, means_hat = m._params(params)
    y_hat, y_lik = m.gaussian_conditional(y, scales_hat, means=means_hat, noise=noise_y)
    m.last_likelihoods = (y_lik, z_lik)
    return synthesis_autograd(m, y_hat)


# --------------------------------------------------------------------------------------------- #
# trainable Bottleneck blocks (stage 2): BatchNorm2d in training mode + ReLU + residual add
# --------------------------------------------------------------------------------------------- #
class _BnActFn(torch.autograd.Function):
    """y = relu?(batch_norm(x; batch statistics) (+ residual)) on bf16 NHWC maps (bn.hip): two passes forward, two backward; the
    ReLU gradient, the gradient of the residual operand and d gamma / d beta come out of the same two passes.  Running statistics
    are updated in place as nn.BatchNorm2d does (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, x_nhwc, gamma, beta, residual, running_mean, running_var, momentum, eps, relu):
        y, mean, rstd = hip.bn_train_fwd(x_nhwc, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps, relu,
                                         residual=residual)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(x_nhwc, y if relu else None, gamma, mean, rstd)
        return y            # (running_mean / running_var: buffers, updated in place by the kernel, not part of the graph)

    @staticmethod
    def backward(ctx, dy):
        x_nhwc, y, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx, dz, dgamma, dbeta = hip.bn_train_bwd(dy, x_nhwc, y, gamma.detach(), mean, rstd, want_dz=ctx.has_res and ctx.relu)
        if ctx.has_res and not ctx.relu:
            dz = dy                                  # (no ReLU: the residual operand's gradient is the incoming one)
        return (dx if ctx.needs_input_grad[0] else None, dgamma.to(gamma.dtype) if ctx.needs_input_grad[1] else None,
                dbeta.to(gamma.dtype) if ctx.needs_input_grad[2] else None, dz if (ctx.has_res and ctx.needs_input_grad[3]) else None,
                None, None, None, None, None)


def bn_module_ok(bn):
    """True if `bn` can take the HIP path: a training-mode nn.BatchNorm2d with f32 affine parameters and running statistics, a
    fixed momentum, C % 8 == 0 and C <= 2048 (decided from the module alone, before anything runs)."""
    return (hip.host_policy.bn_train_hip and type(bn) is torch.nn.BatchNorm2d and bn.training and bn.affine and bn.track_running_stats and
            bn.momentum is not None and bn.weight.dtype == torch.float32 and bn.weight.is_cuda and bn.running_mean is not None and
            bn.running_mean.dtype == torch.float32 and bn.num_features % 8 == 0 and bn.num_features <= 2048)


def _nhwc_bf16(t):
    """NCHW-shaped tensor -> contiguous bf16 NHWC view (a copy only if it is not bf16 channels_last already)."""
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    return t.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


def bn_act(bn, x, relu, residual=None):
    """relu?(bn(x) (+ residual)) for a training-mode nn.BatchNorm2d (bn_module_ok) on NCHW-shaped device tensors; bf16 channels_last
    in (anything else is converted), a bf16 channels_last view out."""
    if x.numel() // x.shape[1] <= 1:      # nn.BatchNorm2d's own refusal (torch.nn.functional._verify_batch_size): variance of one value
        raise ValueError('Expected more than 1 value per channel when training, got input size {}'.format(x.size()))
    with torch.no_grad():
        bn.num_batches_tracked.add_(1)
    res = _nhwc_bf16(residual) if residual is not None else None
    y = _BnActFn.apply(_nhwc_bf16(x), bn.weight, bn.bias, res, bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps),
                       bool(relu))
    return y.permute(0, 3, 1, 2)


def conv_module_ok(conv):
    """True if a TRAINABLE nn.Conv2d can run forward and backward on the library's kernels (`_ConvFn`: implicit-GEMM forward, data
    gradient on the same kernel, weight gradient on conv_wgrad.hip): no bias, no groups, no dilation, channel counts % 8 == 0."""
    return (hip.host_policy.conv_train_hip and type(conv) is torch.nn.Conv2d and conv.bias is None and conv.groups == 1 and
            conv.dilation == (1, 1) and conv.padding_mode == 'zeros' and conv.weight.is_cuda and conv.weight.dtype == torch.float32 and
            conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and not isinstance(conv.padding, str))


def conv_train(conv, x):
    """conv(x) for a trainable nn.Conv2d (conv_module_ok) on an NCHW-shaped device tensor: bf16 operands, f32 accumulation, the f32
    parameter packed to bf16 per call (it changes every step); a bf16 channels_last view out."""
    kh, kw = conv.kernel_size
    order = hip.preferred_k_order(conv.in_channels, kh, kw)
    with torch.no_grad():
        packed = hip.pack_conv_weight(conv.weight, order)
    y = _ConvFn.apply(_nhwc_bf16(x), conv.weight, packed, kh, kw, tuple(conv.stride), tuple(conv.padding), hip.OUT_BF16_NHWC, None, None, order)
    return y.permute(0, 3, 1, 2)

