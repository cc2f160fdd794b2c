"""Frozen ResNet stacks of the distillation step on the HIP kernels: forward AND input gradient.

Stage 1 of the Entropic-Student recipe trains only the bottleneck: the teacher is frozen and runs without gradients, and the
student's `layer2..layer4` are `frozen_modules` in eval mode (reference config
configs/ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_resnet50.yaml:99-139), but the
feature-matching losses sit BEHIND them, so every step needs their forward and the gradient with respect to their input.  Frozen
+ eval means BatchNorm is an affine map: each conv + norm (+ ReLU) (+ residual) is one launch of the inference head
(`head._Conv`), and the data gradient of such a launch is again a convolution -- 1x1: the transposed folded weight; 3x3 stride 1:
the flipped, transposed folded weight, padding k - 1 - p -- so it runs on the SAME window-plane / streaming / weights-in-registers
kernels; strided layers go through `hip.conv2d_dgrad` (stride-parity classes on the tile kernel).  ReLU masks come from the
saved outputs (`hip.relu_bwd`, which also sums the two gradient branches that meet at a block boundary).

`FrozenStackFn` is the autograd node of one stack (one `layerN`): torch autograd adds the loss gradients that arrive at each
stack's output (the per-layer MSE terms of the recipe) before calling its backward.
"""
import torch

from . import hip
from .autograd import MseSink
from .head import ConvSpec, _Conv
from .resnet import Bottleneck


class _Dgrad(object):
    """Input gradient of one folded conv launch."""

    def __init__(self, conv, tag):
        self.stride, self.pad, self.k = conv.stride, conv.pad, conv.k
        self.w_folded = conv.w_folded
        self.as_conv = None
        self._packed = {}      # strided layers: the packed sub-filters of hip.conv2d_dgrad, built at the first call (frozen weight)
        if conv.stride == (1, 1) and conv.dilation == (1, 1):
            wd = conv.w_folded.permute(1, 0, 2, 3).flip(2, 3).contiguous()
            self.as_conv = _Conv(ConvSpec(wd, (1, 1), (self.k[0] - 1 - self.pad[0], self.k[1] - 1 - self.pad[1])), None, tag)
        # a 1x1 stride-2 layer (the downsample of a stage's first block): its input gradient lives on the even pixels only, where it
        # is the stride-1 1x1 conv of g with the transposed weights -- `dense()`; the caller adds it into the other branch in place
        self.as_dense = None
        if tuple(conv.stride) == (2, 2) and tuple(self.k) == (1, 1) and tuple(self.pad) == (0, 0) and conv.dilation == (1, 1):
            wd = conv.w_folded.permute(1, 0, 2, 3).contiguous()
            self.as_dense = _Conv(ConvSpec(wd, (1, 1), (0, 0)), None, tag)

    def dense(self, g):
        return self.as_dense(g, hip.EPI_BIAS)

    def __call__(self, g, in_hw, mask=None, add=None):
        """mask: the saved output of the ReLU in front of this layer's input -- the result is the gradient IN FRONT of that ReLU;
        add (with mask): a second gradient reaching the same tensor (the block's skip path), summed in front of the mask."""
        assert add is None or mask is not None
        if self.as_conv is not None:
            return self.as_conv(g, hip.EPI_BIAS, ep_x=add, ep_mask=mask) if mask is not None else self.as_conv(g, hip.EPI_BIAS)
        gx = hip.conv2d_dgrad(g, self.w_folded, self.stride, self.pad, in_hw, cache=self._packed)
        return gx if mask is None else hip.relu_bwd(gx, mask, add=add)


class FrozenStack(object):
    """One stack of torchvision Bottleneck blocks with every norm layer folded (eval mode, frozen parameters)."""

    def __init__(self, name, layer):
        if not all(isinstance(b, Bottleneck) for b in layer):
            raise hip.Sc2Error('FrozenStack supports stacks of Bottleneck blocks')
        self.name = name
        self.blocks = []
        for bi, blk in enumerate(layer):
            t = '{}.{}'.format(name, bi)
            c1, c2, c3 = _Conv(blk.conv1, blk.bn1, t + '.c1'), _Conv(blk.conv2, blk.bn2, t + '.c2'), _Conv(blk.conv3, blk.bn3, t + '.c3')
            ds = _Conv(blk.downsample[0], blk.downsample[1], t + '.ds') if blk.downsample is not None else None
            self.blocks.append((c1, c2, c3, ds))
        self._dgrads = None

    @staticmethod
    def supported(layer):
        return isinstance(layer, torch.nn.Sequential) and len(layer) > 0 and all(
            isinstance(b, Bottleneck) and all(c.dilation == (1, 1) for c in (b.conv1, b.conv2, b.conv3)) for b in layer) and \
            not any(p.requires_grad for p in layer.parameters()) and not layer.training

    def forward(self, x_nhwc, save=False):
        """bf16 NHWC -> bf16 NHWC; `save`: also the tensors the input gradient needs, per block (input, o1, o2, output)."""
        h, saved = x_nhwc, []
        for c1, c2, c3, ds in self.blocks:
            identity = h if ds is None else ds(h, hip.EPI_BIAS)
            o1 = c1(h, hip.EPI_BIAS_RELU)
            o2 = c2(o1, hip.EPI_BIAS_RELU)
            out = c3(o2, hip.EPI_BIAS_ADD_RELU, ep_x=identity)
            if save:
                saved.append((h, o1, o2, out))
            h = out
        return h, saved

    def _dg(self):
        if self._dgrads is None:      # built on first use: inference-only callers (the teacher) never pay for them
            self._dgrads = [tuple(None if c is None else _Dgrad(c, c.tag + '.dgrad') for c in blk) for blk in self.blocks]
        return self._dgrads

    def backward(self, g_out, saved, mse=None):
        """gradient with respect to the stack's input, given the gradient at its output (bf16 NHWC, contiguous; None: no gradient
        but the MSE term's reaches it).  `mse` = (teacher features laid out like the output, f32 device scale): a feature-matching
        MSE term on the stack's OUTPUT whose gradient 2 scale (out - t) is formed inside the first ReLU-gradient pass."""
        dgs = self._dg()
        g_a, g_b = g_out, None          # the two branches that meet at the current block's output
        masked = False                  # g_a already IS the gradient in front of this block's output ReLU
        for bi in range(len(self.blocks) - 1, -1, -1):
            h, o1, o2, out = saved[bi]
            d1, d2, d3, dds = dgs[bi]
            if masked:
                g = g_a
            elif mse is not None and bi == len(self.blocks) - 1:
                g = hip.relu_bwd_mse(g_a, out, mse[0], mse[1])
            else:
                g = hip.relu_bwd(g_a, out, add=g_b)
            g2 = d3(g, o2.shape[1:3], mask=o2)       # (the ReLU gradients ride in the data-gradient launches' epilogues where
            g1 = d2(g2, o1.shape[1:3], mask=o1)      #  the layer runs on a kernel that takes a mask: head._Conv, ep_mask)
            masked = False
            if dds is None and bi > 0 and d1.as_conv is not None:
                # no downsample: this block's input IS the previous block's output h (saved[bi - 1][3]).  Its gradient in front of
                # that block's ReLU = (conv1's data gradient + g over the skip path) * (h > 0): sum and mask in the launch
                g_a, g_b, masked = d1(g1, h.shape[1:3], mask=h, add=g), None, True
                continue
            g_c1 = d1(g1, h.shape[1:3])
            if dds is not None and dds.as_dense is not None:
                # (was: scatter into a zero-filled map, then a full-size add -- three fills and 1.2 GB of traffic per block)
                sub = dds.dense(g)
                g_c1[:, ::2, ::2][:, :sub.shape[1], :sub.shape[2]].add_(sub)
                g_a, g_b = g_c1, None
            elif dds is not None:
                g_a, g_b = g_c1 + dds(g, h.shape[1:3]), None
            else:
                g_a, g_b = g_c1, g
        return g_a if g_b is None else g_a + g_b


class FrozenStackFn(torch.autograd.Function):
    """out = stack(x) for a bf16 channels_last NCHW tensor x; backward = FrozenStack.backward."""

    @staticmethod
    def forward(ctx, x, stack):
        x_nhwc = x.permute(0, 2, 3, 1)
        if not x_nhwc.is_contiguous():
            x_nhwc = x_nhwc.contiguous()
        out, saved = stack.forward(x_nhwc, save=x.requires_grad)
        ctx.stack, ctx.saved = stack, saved
        # MSE terms on this output hand their (target, scale) to this node instead of a gradient tensor (MseSumFn.backward; autograd
        # runs every consumer's backward before the producer's): the sink travels on the output tensor
        ctx.mse_sink = MseSink()
        ctx.set_materialize_grads(False)       # no other gradient: None, not a zero-filled map
        res = out.permute(0, 3, 1, 2)
        res._sc2_mse_sink = ctx.mse_sink
        return res

    @staticmethod
    def backward(ctx, gy):
        sink = ctx.mse_sink.drain()      # (the MSE nodes keep the list itself: emptied, not replaced; this pass's entries only)
        if gy is None and not sink:
            ctx.saved = None
            return None, None
        g = None
        if gy is not None:
            g = gy.permute(0, 2, 3, 1)
            if g.dtype != torch.bfloat16 or not g.is_contiguous():
                g = g.to(torch.bfloat16).contiguous()
        mse = None
        out = ctx.saved[-1][3]
        for y, scale in sink:            # (one term per output in the recipes; further ones as gradient tensors)
            t = y.permute(0, 2, 3, 1)
            if mse is None and hip._same_dense_bf16(out, t):
                mse = (t, scale)
            else:
                extra = hip.mse_grad(out, t.contiguous(), scale)
                g = extra if g is None else g + extra
        gx = ctx.stack.backward(g, ctx.saved, mse=mse)
        ctx.saved = None
        return gx.permute(0, 3, 1, 2), None


class MseSumFn(torch.autograd.Function):
    """nn.MSELoss(reduction='sum' | 'mean')(x, y) for bf16 tensors of one dense layout: one pass forward (f32 accumulation),
    one pass backward; y is a target (no gradient)."""

    @staticmethod
    def forward(ctx, x, y, mean, sink=None, wants_x=False):
        ctx.save_for_backward(x, y)
        ctx.div = float(x.numel()) if mean else 1.0
        ctx.sink, ctx.wants_x = sink, wants_x
        return hip.mse_sum(x, y) / ctx.div

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        scale = (g.float() / ctx.div).reshape(1).contiguous()
        if ctx.sink is not None:
            # x is the output of a frozen stack (FrozenStackFn): its backward forms 2 scale (x - y) inside its first ReLU-gradient pass
            # (sc2_relu_bwd_mse_bf16) -- no gradient tensor, no add
            ctx.sink.put((x, y, scale) if ctx.wants_x else (y, scale))      # (a conv node does not keep its output: it gets x too)
            return None, None, None, None, None
        return hip.mse_grad(x, y, scale), None, None, None, None


def mse_fast_path(loss_module, x, y):
    """The HIP form of `loss_module(x.float(), y.float())` when it applies (MSELoss sum / mean on two bf16 device tensors of one
    shape and dense layout, gradient only to x), else None."""
    if type(loss_module) is not torch.nn.MSELoss or loss_module.reduction not in ('sum', 'mean'):
        return None
    if not hip._same_dense_bf16(x, y) or y.requires_grad:
        return None
    sink = getattr(x, '_sc2_mse_sink', None) if hip.host_policy.mse_fused and x.requires_grad else None
    if sink is not None and (x.retains_grad or getattr(x, '_backward_hooks', None)):
        sink = None       # someone reads d loss / d x off the tensor itself: it must carry the MSE term (MseSink's restriction)
    return MseSumFn.apply(x, y.detach(), loss_module.reduction == 'mean', sink, bool(getattr(x, '_sc2_mse_sink_wants_x', False)))
