"""Splittable backbones: host-side mirror of sc2bench/models/backbone.py for the ResNet path
(UpdatableBackbone :47-75, FeatureExtractionBackbone :90-172, SplittableResNet :175-276, splittable_resnet :658-698, registries :15-44,
get_backbone :894-909).

Forward order is the reference's: pre_transform -> bottleneck (encode/analyze/decode when updated and in
eval mode, else bottleneck.forward) -> layer2 -> layer3 -> layer4 -> avgpool -> flatten -> fc
(backbone.py:225-254).  ``compute_dtype='bf16'`` runs the task head in bf16 channels_last and takes the
decoder's bf16 NHWC output without a copy.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import hip
from .analysis import AnalyzableModule
from .entropy import CompressionModel
from .layer import FPBasedResNetBottleneck, get_layer
from .resnet import RESNET_FUNC_DICT, FrozenBatchNorm2d

BACKBONE_CLASS_DICT = dict()
BACKBONE_FUNC_DICT = dict()
MODEL_DICT = dict()  # the slice of torchdistill's model registry this package feeds


def register_backbone_class(cls):
    BACKBONE_CLASS_DICT[cls.__name__] = cls
    MODEL_DICT[cls.__name__] = cls
    return cls


def register_backbone_func(func):
    BACKBONE_FUNC_DICT[func.__name__] = func
    MODEL_DICT[func.__name__] = func
    return func


class UpdatableBackbone(AnalyzableModule):
    """Base class of backbones that carry an updatable (entropy-coded) bottleneck."""

    def __init__(self, analyzer_configs=None):
        super().__init__(analyzer_configs)
        self.bottleneck_updated = False

    def forward(self, *args, **kwargs):
        raise NotImplementedError()

    def update(self, **kwargs):
        raise NotImplementedError()

    def get_aux_module(self, **kwargs):
        raise NotImplementedError()

    def _drop_eval_graphs(self):
        """forget captured HIP graphs of the eval forward (graphs.py): their baked-in addresses die with a change of storage / mode"""
        self.__dict__.pop('_eval_graphs', None)


def check_if_updatable(model):
    return isinstance(model, UpdatableBackbone)


class FeatureExtractionBackbone(UpdatableBackbone):
    """Runs the named children of `model` in order and returns the outputs of the ones listed in `return_layer_dict`
    (sc2bench/models/backbone.py:90-172; the body of the detection and segmentation models).  The child named by
    `analyzable_layer_key` (the bottleneck) goes through encode -> analyze -> decode once the model is updated and in
    eval mode; children after the last returned layer are dropped.

    `set_compute_dtype('bf16')` (eval): the bottleneck hands bf16 NHWC features on, and every following child that is a
    stack of torchvision Bottleneck blocks without dilation runs on the fused conv+norm kernels of `head.HipHead`
    (FrozenBatchNorm2d and BatchNorm2d both fold); dilated stacks (DeepLab's layer3 / layer4) stay torch modules in
    bf16 channels_last.
    """

    def __init__(self, model, return_layer_dict, analyzer_configs, analyzes_after_compress=False,
                 analyzable_layer_key=None):
        children = OrderedDict(model.named_children())
        if not set(return_layer_dict).issubset(children.keys()):
            raise ValueError('return_layer_dict are not present in model')
        super().__init__(analyzer_configs)
        wanted = {str(k) for k in return_layer_dict}
        for name, module in children.items():
            self.add_module(name, module)
            wanted.discard(name)
            if not wanted:        # everything to be returned has been produced: the rest of the model is not needed
                break
        self.return_layer_dict = return_layer_dict
        self.analyzable_layer_key = analyzable_layer_key
        self.analyzes_after_compress = analyzes_after_compress
        self.compute_dtype = 'f32'
        self._hip_layers = dict()

    def set_compute_dtype(self, dtype):
        assert dtype in ('f32', 'bf16')
        self._drop_eval_graphs()
        self.compute_dtype = dtype
        for name, module in self.named_children():
            if name == self.analyzable_layer_key:
                if hasattr(module, 'output_format'):
                    module.output_format = 'bf16_nhwc' if dtype == 'bf16' else 'f32_nchw'
            elif dtype == 'bf16':
                module.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
            else:
                module.to(dtype=torch.float32)
        return self

    def _hip_layer(self, name, module):
        """HipHead of one child, or None if the child is not a stack of Bottleneck blocks (dilated 3x3 layers -- DeepLab's
        layer3 / layer4 -- run as d * d undilated launches on the phase grids, `head._Conv._dilated`)."""
        from .head import HipHead
        from .resnet import Bottleneck
        if not (isinstance(module, nn.Sequential) and len(module) > 0 and all(isinstance(b, Bottleneck) for b in module)):
            return None
        if any(c.dilation != (1, 1) and not (c.kernel_size == (1, 1) or (c.kernel_size == (3, 3) and c.stride == (1, 1) and
                                                                         c.padding == c.dilation and c.dilation[0] == c.dilation[1]))
               for b in module for c in (b.conv1, b.conv2, b.conv3)):
            return None
        key = tuple(t._version for t in list(module.parameters()) + list(module.buffers()))
        cached = self._hip_layers.get(name)
        if cached is None or cached[0] != key:
            cached = (key, HipHead([(name, module)], None))
            self._hip_layers[name] = cached
        return cached[1]

    def _run_child(self, name, module, x):
        if self.compute_dtype == 'bf16' and not self.training and x.is_cuda and x.dtype == torch.bfloat16:
            head = self._hip_layer(name, module)
            if head is not None:
                return head.forward(x.permute(0, 2, 3, 1).contiguous(), with_pool=False)
        return module(x)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.named_children():
            if name == self.analyzable_layer_key and self.bottleneck_updated and not self.training:
                compressed = module.encode(x)
                if self.analyzes_after_compress:
                    self.analyze(compressed)
                x = module.decode(**compressed)
            else:
                x = self._run_child(name, module, x)
            if name in self.return_layer_dict:
                out[self.return_layer_dict[name]] = x
        return out

    # ---- the updated eval forward in stages (pipeline.StagePipeline): children in front of the bottleneck + its front stage,
    #      the coder, then its decoder + the remaining children
    @property
    def stage_front_takes_out(self):
        key = self.analyzable_layer_key
        first = next(iter(self._modules), None)
        return key is not None and first == key and bool(getattr(self._modules[key], 'stage_front_takes_out', False))

    @property
    def stage_coder_kwargs(self):
        key = self.analyzable_layer_key
        return dict(getattr(self._modules.get(key), 'stage_coder_kwargs', {})) if key is not None else {}

    def stages_ready(self):
        key = self.analyzable_layer_key
        return (key is not None and key in self._modules and self.bottleneck_updated and not self.training and
                hasattr(self._modules[key], 'stage_front') and next(self.parameters()).is_cuda)

    def stage_front(self, x, out=None):
        for name, module in self.named_children():
            if name == self.analyzable_layer_key:
                return module.stage_front(x, out=out) if out is not None else module.stage_front(x)
            x = self._run_child(name, module, x)
        raise KeyError('`analyzable_layer_key` ({}) does not exist'.format(self.analyzable_layer_key))

    def stage_coder(self, payload, meta, **kwargs):
        return self._modules[self.analyzable_layer_key].stage_coder(payload, meta, **kwargs)

    def stage_back(self, decoded, meta):
        out = OrderedDict()
        seen = False
        x = None
        for name, module in self.named_children():
            if name == self.analyzable_layer_key:
                x = module.synthesis_nhwc(module.stage_decode(decoded, meta))
                seen = True
            elif not seen:
                continue        # (ran in stage_front; a layer in front of the bottleneck is never a returned one in the reference's configs)
            else:
                x = self._run_child(name, module, x)
            if name in self.return_layer_dict:
                out[self.return_layer_dict[name]] = x
        return out

    def check_if_updatable(self):
        key = self.analyzable_layer_key
        return key is not None and key in self._modules and isinstance(self._modules[key], CompressionModel)

    def update(self):
        if self.analyzable_layer_key is None:
            return
        if self.analyzable_layer_key not in self._modules:
            raise KeyError('`analyzable_layer_key` ({}) does not exist in {}'.format(self.analyzable_layer_key, self))
        self._modules[self.analyzable_layer_key].update()
        self.bottleneck_updated = True

    def get_aux_module(self, **kwargs):
        return self._modules[self.analyzable_layer_key] if self.check_if_updatable() else None


register_backbone_class(FeatureExtractionBackbone)


class SplittableResNet(UpdatableBackbone):
    """ResNet with its stem and layer1 replaced by a neural encoder / entropy bottleneck / decoder.

    :param bottleneck_layer: bottleneck module (encoder + entropy bottleneck + decoder)
    :param resnet_model: ResNet to take layer2..fc from
    :param inplanes: ResNet inplanes or None
    :param skips_avgpool: drop avgpool (and everything after)
    :param skips_fc: drop fc
    :param pre_transform: optional module applied to the input
    :param analysis_config: {'analyzes_after_compress': bool, 'analyzer_configs': [...]}
    :param short_module_names: which of layer2/3/4 to keep
    """

    def __init__(self, bottleneck_layer, resnet_model, inplanes=None, skips_avgpool=True, skips_fc=True,
                 pre_transform=None, analysis_config=None, short_module_names=None):
        if analysis_config is None:
            analysis_config = dict()
        if short_module_names is None:
            short_module_name_set = {'layer2', 'layer3', 'layer4'}
        else:
            short_module_name_set = set(short_module_names)
        super().__init__(analysis_config.get('analyzer_configs', list()))
        self.pre_transform = pre_transform
        self.analyzes_after_compress = analysis_config.get('analyzes_after_compress', False)
        self.bottleneck_layer = bottleneck_layer
        self.layer2 = resnet_model.layer2 if 'layer2' in short_module_name_set else None
        self.layer3 = resnet_model.layer3 if 'layer3' in short_module_name_set else None
        self.layer4 = resnet_model.layer4 if 'layer4' in short_module_name_set else None
        self.avgpool = None if skips_avgpool \
            else resnet_model.global_pool if hasattr(resnet_model, 'global_pool') else resnet_model.avgpool
        self.fc = None if skips_fc else resnet_model.fc
        self.inplanes = resnet_model.inplanes if inplanes is None else inplanes
        self.compute_dtype = 'f32'
        self.use_hip_head = True     # bf16 eval: run layer2..fc on the library's fused conv kernel (head.py)
        self._hip_head = None
        self._hip_head_key = None

    def set_compute_dtype(self, dtype):
        """'f32' (reference dtype) or 'bf16' (task head in bf16 channels_last, decoder output zero-copy)."""
        assert dtype in ('f32', 'bf16')
        self._drop_eval_graphs()
        self.compute_dtype = dtype
        head = [m for m in (self.layer2, self.layer3, self.layer4, self.avgpool, self.fc) if m is not None]
        for m in head:
            if dtype == 'bf16':
                m.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
            else:
                m.to(dtype=torch.float32)
        if hasattr(self.bottleneck_layer, 'output_format'):
            self.bottleneck_layer.output_format = 'bf16_nhwc' if dtype == 'bf16' else 'f32_nchw'
        return self

    def set_encoder_precision(self, precision):
        """'f32': the bottleneck's analysis transform with f32 operands, so that symbols / byte streams / bpp are the f32
        reference path's (FPBasedResNetBottleneck.set_encoder_precision); 'bf16': the fast default."""
        self._drop_eval_graphs()
        self.bottleneck_layer.set_encoder_precision(precision)
        return self

    def _hip_head_for_eval(self):
        """Folded conv+BN(+ReLU)(+residual) head for bf16 eval; rebuilt when a parameter changes."""
        mods = [m for m in (self.layer2, self.layer3, self.layer4, self.fc) if m is not None]
        key = tuple(p._version for m in mods for p in m.parameters()) + \
            tuple(b._version for m in mods for b in m.buffers()) + (str(next(mods[0].parameters()).device),)
        if self._hip_head is None or self._hip_head_key != key:
            from .head import HipHead
            layers = [(i + 2, m) for i, m in enumerate((self.layer2, self.layer3, self.layer4)) if m is not None]
            self._hip_head = HipHead(layers, self.fc if self.avgpool is not None else None)
            self._hip_head_key = key
        return self._hip_head

    def head(self, x):
        if (self.compute_dtype == 'bf16' and self.use_hip_head and not self.training and x.is_cuda
                and x.dtype == torch.bfloat16 and self.layer2 is not None):
            x_nhwc = x.permute(0, 2, 3, 1).contiguous()   # a view when x is channels_last (the decoder's output)
            return self._hip_head_for_eval().forward(x_nhwc, with_pool=self.avgpool is not None)
        if self.layer2 is not None:
            x = self.layer2(x)
        if self.layer3 is not None:
            x = self.layer3(x)
        if self.layer4 is not None:
            x = self.layer4(x)
        if self.avgpool is None:
            return x
        x = self.avgpool(x)
        if self.fc is None:
            return x
        x = torch.flatten(x, 1)
        return self.fc(x)

    # ---- HIP-graph replay of the updated eval forward at the reference's evaluation batch size (graphs.py)
    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .half(): storage moves, captured addresses die
        self._drop_eval_graphs()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        if bool(mode) != self.training:
            self._drop_eval_graphs()
        return super().train(mode)

    def _eval_graphs_for(self, x):
        """graphs.EvalGraphs for this input, or None: bf16 eval on the HIP head, an FP bottleneck on the bf16 encoder whose
        tables are built, a batch the host range coder takes (`host_policy.eval_graph_max_batch`, default: batch size 1 only)."""
        if not (hip.host_policy.eval_graphs and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and
                0 < x.shape[0] <= min(hip.host_policy.eval_graph_max_batch, hip.host_coder_max_streams()) and
                self.compute_dtype == 'bf16' and self.use_hip_head and self.layer2 is not None and not torch.is_grad_enabled() and
                type(self.bottleneck_layer) is FPBasedResNetBottleneck and
                getattr(self.bottleneck_layer, 'encoder_precision', 'bf16') == 'bf16' and
                getattr(self.bottleneck_layer, 'output_format', '') == 'bf16_nhwc' and x.dtype == torch.float32 and
                not torch.cuda.is_current_stream_capturing()):
            return None
        from .graphs import graphs_for
        return graphs_for(self, x)

    def _forward_graphed(self, g, x):
        """forward() of the updated eval model on captured graphs: the same encode -> {'strings', 'shape'} -> analyzers -> decode ->
        head sequence, the device halves replayed instead of launched kernel by kernel."""
        eb = self.bottleneck_layer.entropy_bottleneck
        tables = eb._host_tables()
        sym_h = g.symbols(x)
        hw = g.latent_shape[0] * g.latent_shape[1]
        strings, st = hip.rans_encode_host(tables, sym_h, index_div=hw)
        from .entropy import _raise_on_status, _status_or
        if _status_or(st) & 1:     # a row overflowed 2 B / symbol: redo with the proven upper bound (compress_symbols does the same)
            strings, st = hip.rans_encode_host(tables, sym_h, index_div=hw, out_stride=hip.rans_max_bytes(sym_h.shape[1]))
        _raise_on_status(torch.from_numpy(st), 'EntropyBottleneck.compress')
        compressed = {'strings': [strings], 'shape': torch.Size(g.latent_shape)}
        if self.analyzes_after_compress:
            self.analyze(compressed)
        dec_h, st_h = hip.rans_decode_host(tables, compressed['strings'][0], sym_h.shape[1], index_div=hw)
        _raise_on_status(st_h, 'EntropyBottleneck.decompress')
        return g.decode_head(dec_h)

    def forward(self, x):
        if self.pre_transform is not None:
            x = self.pre_transform(x)
        if self.bottleneck_updated and not self.training:
            g = self._eval_graphs_for(x)
            if g is not None:
                return self._forward_graphed(g, x)
            x = self.bottleneck_layer.encode(x)
            if self.analyzes_after_compress:
                self.analyze(x)
            x = self.bottleneck_layer.decode(**x)
        else:
            # the bottleneck chooses its own arithmetic (bf16 MFMA operands, f32 accumulation, f32 entropy model): a caller's
            # autocast region (a training tail in reduced precision) must not re-type the torch ops between its kernels
            with torch.autocast(device_type=x.device.type if x.device.type in ('cuda', 'cpu') else 'cuda', enabled=False):
                x = self.bottleneck_layer(x)
        return self.head(x)

    def forward_device(self, x):
        """Eval-mode forward that keeps the entropy-coded streams on the device (no host round trip):
        returns (head output, nbytes i32 [N]).  Same arithmetic as forward() in updated+eval mode."""
        if self.pre_transform is not None:
            x = self.pre_transform(x)
        buf, off, nb, st, shape = self.bottleneck_layer.encode_device(x)
        x = self.bottleneck_layer.decode_device(buf, off, nb, shape)
        return self.head(x), nb, st

    # ---- the same eval forward cut into three stages, so that a caller (pipeline.StagePipeline) can run the serial range coder
    #      on its own HIP stream(s) while the MFMA streams work on neighbouring batches
    @property
    def stage_front_takes_out(self):
        return bool(getattr(self.bottleneck_layer, 'stage_front_takes_out', False)) and hasattr(self.bottleneck_layer, 'stage_front')

    @property
    def stage_coder_kwargs(self):
        return dict(getattr(self.bottleneck_layer, 'stage_coder_kwargs', {}))

    def stages_ready(self):
        return (self.bottleneck_updated and not self.training and hasattr(self.bottleneck_layer, 'stage_front') and
                next(self.parameters()).is_cuda)

    def stage_front(self, x, out=None):
        """encoder + quantisation: -> (payload, meta) of the bottleneck's `stage_front` (FP bottleneck: symbols int32 [N, C*h*w]
        and (h, w); `out`: a row block of the buffer one range-coder launch will read).  A bottleneck without stages: analysis +
        its entropy bottleneck's symbols."""
        if self.pre_transform is not None:
            x = self.pre_transform(x)
        bl = self.bottleneck_layer
        if hasattr(bl, 'stage_front'):
            return bl.stage_front(x, out=out) if out is not None else bl.stage_front(x)
        latent = bl.analysis(x)
        sym = bl.entropy_bottleneck.symbols_device(latent)
        if out is not None:
            out.copy_(sym)
            sym = out
        return sym, tuple(latent.shape[-2:])

    def stage_coder(self, payload, meta, **kwargs):
        """rANS encode to byte streams, then decode them: -> (decoded, nbytes [N], status [N]) of the bottleneck's `stage_coder`
        (FP bottleneck, `dequantized=True`: the dequantised latent as bf16 NHWC straight from the coder's last pass)."""
        bl = self.bottleneck_layer
        if hasattr(bl, 'stage_coder'):
            return bl.stage_coder(payload, meta, **kwargs)
        return FPBasedResNetBottleneck.stage_coder(bl, payload, meta, **kwargs)

    def stage_coder_host(self, payload, meta, **kwargs):
        """the coder stage on the host thread pool, where the bottleneck offers it (FP bottleneck: `stage_coder_host`)"""
        return self.bottleneck_layer.stage_coder_host(payload, meta, **kwargs)

    @property
    def has_stage_coder_host(self):
        return type(self.bottleneck_layer) is FPBasedResNetBottleneck

    def decode_head(self, y_hat_nhwc):
        """decoder + task head on a dequantised bf16 NHWC latent.  In bf16 eval with the HIP head, the decoder's last conv takes
        layer2.0's conv1 and downsample with it (one launch, `FPBasedResNetBottleneck.synthesis_nhwc_tail`): the 256-channel
        56 x 56 map between bottleneck and head is never written."""
        bl = self.bottleneck_layer
        if (self.compute_dtype == 'bf16' and self.use_hip_head and not self.training and self.layer2 is not None
                and hasattr(bl, 'synthesis_nhwc_tail') and getattr(bl, 'output_format', '') == 'bf16_nhwc'):
            head = self._hip_head_for_eval()
            res = bl.synthesis_nhwc_tail(y_hat_nhwc, head)
            if res is not None:
                feats, pre = res
                if pre is not None:
                    return head.forward(None, with_pool=self.avgpool is not None, pre=pre)
                return self.head(feats)
        return self.head(bl.synthesis_nhwc(y_hat_nhwc))

    def stage_decoder(self, decoded, meta):
        """dequantise + decoder: what the coder stage returned -> features (the MFMA-bound half of the back stage)."""
        bl = self.bottleneck_layer
        y_hat = bl.stage_decode(decoded, meta) if hasattr(bl, 'stage_decode') else FPBasedResNetBottleneck.stage_decode(bl, decoded, meta)
        return bl.synthesis_nhwc(y_hat)

    def stage_back(self, decoded, meta, after_decoder=None):
        """dequantise + decoder + task head.  `after_decoder()` is called between the two."""
        if after_decoder is None:
            bl = self.bottleneck_layer
            y_hat = bl.stage_decode(decoded, meta) if hasattr(bl, 'stage_decode') else FPBasedResNetBottleneck.stage_decode(bl, decoded, meta)
            return self.decode_head(y_hat)
        feats = self.stage_decoder(decoded, meta)
        after_decoder()
        return self.head(feats)

    def update(self):
        self._drop_eval_graphs()
        self.bottleneck_layer.update()
        self.bottleneck_updated = True

    def load_state_dict(self, state_dict, **kwargs):
        """Loads everything but `bottleneck_layer.*` non-strictly, then the bottleneck through its own loader
        (which resizes the CDF buffers).  Like the reference, this pops the bottleneck keys from the passed dict."""
        self._drop_eval_graphs()
        entropy_bottleneck_state_dict = OrderedDict()
        for key in list(state_dict.keys()):
            if key.startswith('bottleneck_layer.'):
                entropy_bottleneck_state_dict[key.replace('bottleneck_layer.', '', 1)] = state_dict.pop(key)
        super().load_state_dict(state_dict, strict=False)
        self.bottleneck_layer.load_state_dict(entropy_bottleneck_state_dict)

    def get_aux_module(self, **kwargs):
        return self.bottleneck_layer if isinstance(self.bottleneck_layer, CompressionModel) else None


register_backbone_class(SplittableResNet)


@register_backbone_func
def splittable_resnet(bottleneck_config, resnet_name='resnet50', inplanes=None, skips_avgpool=True, skips_fc=True,
                      pre_transform=None, analysis_config=None, org_model_ckpt_file_path_or_url=None,
                      org_ckpt_strict=True, short_module_names=None, **resnet_kwargs):
    """Builds a SplittableResNet from a bottleneck config {'key', 'kwargs'} and a ResNet name
    (same signature as backbone.py:659-698)."""
    bottleneck_layer = get_layer(bottleneck_config['key'], **bottleneck_config['kwargs'])
    if resnet_name not in RESNET_FUNC_DICT:
        raise KeyError('resnet_name `{}` is not available (have {})'.format(resnet_name, sorted(RESNET_FUNC_DICT)))
    if resnet_kwargs.pop('norm_layer', '') == 'FrozenBatchNorm2d':
        resnet_model = RESNET_FUNC_DICT[resnet_name](norm_layer=FrozenBatchNorm2d, **resnet_kwargs)
    else:
        resnet_model = RESNET_FUNC_DICT[resnet_name](**resnet_kwargs)
    if org_model_ckpt_file_path_or_url is not None:
        from .ckpt import load_ckpt
        load_ckpt(org_model_ckpt_file_path_or_url, model=resnet_model, strict=org_ckpt_strict)
    return SplittableResNet(bottleneck_layer, resnet_model, inplanes, skips_avgpool, skips_fc,
                            pre_transform, analysis_config, short_module_names=short_module_names)


def get_backbone(cls_or_func_name, **kwargs):
    if cls_or_func_name in BACKBONE_CLASS_DICT:
        return BACKBONE_CLASS_DICT[cls_or_func_name](**kwargs)
    elif cls_or_func_name in BACKBONE_FUNC_DICT:
        return BACKBONE_FUNC_DICT[cls_or_func_name](**kwargs)
    return None
