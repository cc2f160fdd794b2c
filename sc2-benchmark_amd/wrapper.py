"""Model wrappers of the reference's baselines (sc2bench/models/wrapper.py), built against the same contract:
registry `WRAPPER_CLASS_DICT`, constructor keywords as the YAML files pass them, `forward` semantics, analyzers.

* `NeuralInputCompressionClassifier` (wrapper.py:80-135): pre_transform -> compression_model.compress -> analyze the
  compressed object -> decompress (`x_hat`) -> post_transform -> classifier.  BASELINE config 3 drives the HIP-backed
  `compression.FactorizedPrior` through it.
* `CodecFeatureCompressionClassifier` (wrapper.py:138-193): classifier split into encoder / decoder / classifier by
  `sequential` module lists; each sample's feature map goes through a codec transform (config 1: `PILTensorModule`).
* `CodecInputCompressionClassifier` (wrapper.py:28-77), `EntropicClassifier` (:196-265), `SplitClassifier` (:268-322).

Reference behaviour kept on purpose: per-sample results are joined with `torch.hstack`, i.e. along dim 1 -- the
reference evaluates these baselines at test batch size 1 (README.md:100-108), where that equals a batch of one.
"""
from collections import OrderedDict

import torch
from torch import nn

from .analysis import AnalyzableModule
from .backbone import UpdatableBackbone
from .layer import EntropyBottleneckLayer
from .training import redesign_model

WRAPPER_CLASS_DICT = dict()


def register_wrapper_class(cls):
    WRAPPER_CLASS_DICT[cls.__name__] = cls
    return cls


def _section(model, config):
    """`{sequential: [...]}` -> the named children of `model` in a row; `{ignored: True}` -> identity."""
    config = config or dict()
    if config.get('ignored', False):
        return nn.Identity()
    return redesign_model(model, config.get('sequential'))


def _analyzer_configs(analysis_config):
    return (analysis_config or dict()).get('analyzer_configs', list())


class _PerSampleCodec(object):
    """codec -> analyze(file size) -> post_transform for every sample of a batch, joined as the reference joins them."""

    def _through_codec(self, samples):
        outputs = []
        for sample in samples:
            if self.codec_encoder_decoder is not None:
                sample, file_size = self.codec_encoder_decoder(sample)
                if not self.training:
                    self.analyze(file_size)
            if self.post_transform is not None:
                sample = self.post_transform(sample)
            outputs.append(sample.unsqueeze(0))
        return torch.hstack(outputs).to(self.device)


@register_wrapper_class
class CodecInputCompressionClassifier(AnalyzableModule, _PerSampleCodec):
    """Codec (JPEG / WebP / BPG ...) on the input image, then a classifier.  `x`: list of PIL images."""

    def __init__(self, classification_model, device, codec_encoder_decoder=None, post_transform=None,
                 analysis_config=None, **kwargs):
        super().__init__(_analyzer_configs(analysis_config))
        self.codec_encoder_decoder = codec_encoder_decoder
        self.device = device
        self.classification_model = classification_model
        self.post_transform = post_transform

    def forward(self, x):
        return self.classification_model(self._through_codec(x))


@register_wrapper_class
class NeuralInputCompressionClassifier(AnalyzableModule):
    """Learned image codec on the input, then a classifier."""

    def __init__(self, classification_model, pre_transform=None, compression_model=None,
                 uses_cpu4compression_model=False, post_transform=None, analysis_config=None, **kwargs):
        analysis_config = analysis_config or dict()
        super().__init__(_analyzer_configs(analysis_config))
        self.analyzes_after_pre_transform = analysis_config.get('analyzes_after_pre_transform', False)
        self.analyzes_after_compress = analysis_config.get('analyzes_after_compress', False)
        self.pre_transform = pre_transform
        self.compression_model = compression_model
        self.uses_cpu4compression_model = uses_cpu4compression_model
        self.classification_model = classification_model
        self.post_transform = post_transform

    def use_cpu4compression(self):
        """The reference can park the codec on the CPU (wrapper.py:113-119); the HIP-backed codec has no CPU path, so a
        model that asks for it is told, not silently moved."""
        if self.uses_cpu4compression_model and self.compression_model is not None:
            from . import hip
            raise hip.Sc2Error('uses_cpu4compression_model: the compression model of this build runs on a HIP device only')

    # ---- the same forward in stages (pipeline.StagePipeline): the codec's serial range coder on its own HIP stream
    stage_front_takes_out = False
    stage_coder_kwargs = {'dequantized': True}

    def stages_ready(self):
        cm = self.compression_model
        return (cm is not None and hasattr(cm, 'stage_front') and not self.training and next(cm.parameters()).is_cuda and
                cm.entropy_bottleneck._quantized_cdf.numel() > 0)

    def stage_front(self, x, out=None):
        if self.pre_transform is not None:
            x = self.pre_transform(x)
        return self.compression_model.stage_front(x)

    def stage_coder(self, payload, meta, **kwargs):
        return self.compression_model.stage_coder(payload, meta, **kwargs)

    def stage_back(self, decoded, meta):
        x = self.compression_model.stage_back(decoded, meta)
        if self.post_transform is not None:
            x = self.post_transform(x)
        return self._classify(x)

    def set_compute_dtype(self, dtype):
        """'f32' (the reference's dtype; default): the classifier's own torch modules.  'bf16': a torchvision-layout ResNet
        classifier in eval mode runs on the library's fused conv + norm kernels (head.HipResNet), as the Entropic-Student
        model's head does; any other classifier keeps its modules."""
        assert dtype in ('f32', 'bf16')
        self.compute_dtype = dtype
        return self

    def _classify(self, x):
        clf = self.classification_model
        if getattr(self, 'compute_dtype', 'f32') == 'bf16' and x.is_cuda and not self.training:
            from .head import HipResNet
            if HipResNet.supported(clf):
                ent = self.__dict__.get('_hip_clf')
                if ent is None or ent.key != HipResNet.version_key(clf):
                    ent = HipResNet(clf)
                    self.__dict__['_hip_clf'] = ent
                return ent.forward(x)
        return clf(x)

    def forward(self, x):
        if self.pre_transform is not None:
            x = self.pre_transform(x)
            if self.analyzes_after_pre_transform and not self.training:
                self.analyze(x)
        if self.compression_model is not None:
            compressed_obj = self.compression_model.compress(x)
            if self.analyzes_after_compress and not self.training:
                self.analyze(compressed_obj)
            x = self.compression_model.decompress(**compressed_obj)
            if isinstance(x, dict):
                x = x['x_hat']
        if self.post_transform is not None:
            x = self.post_transform(x)
        return self._classify(x)


@register_wrapper_class
class CodecFeatureCompressionClassifier(AnalyzableModule, _PerSampleCodec):
    """Codec on an intermediate feature map of the classifier (the jpeg-resnet50 feature-compression baseline)."""

    def __init__(self, classification_model, device, encoder_config=None, codec_encoder_decoder=None,
                 decoder_config=None, classifier_config=None, post_transform=None, analysis_config=None, **kwargs):
        super().__init__(_analyzer_configs(analysis_config))
        self.codec_encoder_decoder = codec_encoder_decoder
        self.device = device
        self.encoder = _section(classification_model, encoder_config)
        self.decoder = _section(classification_model, decoder_config)
        self.classifier = _section(classification_model, classifier_config)
        self.post_transform = post_transform

    def forward(self, x):
        x = self._through_codec(self.encoder(x))
        x = self.decoder(x)
        return self.classifier(torch.flatten(x, 1))


class _SplitBase(UpdatableBackbone):
    def __init__(self, classification_model, encoder_config, decoder_config, classifier_config, analysis_config):
        analysis_config = analysis_config or dict()
        super().__init__(_analyzer_configs(analysis_config))
        self.analyzes_after_compress = analysis_config.get('analyzes_after_compress', False)
        self.encoder = _section(classification_model, encoder_config)
        self.decoder = _section(classification_model, decoder_config)
        self.classifier = _section(classification_model, classifier_config)

    def _tail(self, x):
        x = self.decoder(x)
        return self.classifier(torch.flatten(x, 1))


@register_wrapper_class
class EntropicClassifier(_SplitBase):
    """An `EntropyBottleneckLayer` between two halves of a classifier."""

    def __init__(self, classification_model, encoder_config, compression_model_kwargs, decoder_config,
                 classifier_config, analysis_config=None, **kwargs):
        super().__init__(classification_model, encoder_config, decoder_config, classifier_config, analysis_config)
        self.entropy_bottleneck = EntropyBottleneckLayer(**compression_model_kwargs)

    def forward(self, x):
        x = self.encoder(x)
        if self.bottleneck_updated and not self.training:
            x = self.entropy_bottleneck.compress(x)
            if self.analyzes_after_compress:
                self.analyze(x)
            x = self.entropy_bottleneck.decompress(**x)
        else:
            x, _ = self.entropy_bottleneck(x)
        return self._tail(x)

    def update(self):
        self.entropy_bottleneck.update()
        self.bottleneck_updated = True

    def load_state_dict(self, state_dict, **kwargs):
        own = OrderedDict((k[len('entropy_bottleneck.'):], state_dict.pop(k)) for k in list(state_dict.keys())
                          if k.startswith('entropy_bottleneck.'))
        super().load_state_dict(state_dict, strict=False)
        self.entropy_bottleneck.load_state_dict(own)

    def get_aux_module(self, **kwargs):
        return self.entropy_bottleneck


@register_wrapper_class
class SplitClassifier(_SplitBase):
    """A classifier split in two with an optional (de)compressor transform pair at the cut."""

    def __init__(self, classification_model, encoder_config, decoder_config, classifier_config,
                 compressor_transform=None, decompressor_transform=None, analysis_config=None, **kwargs):
        super().__init__(classification_model, encoder_config, decoder_config, classifier_config, analysis_config)
        self.compressor = compressor_transform
        self.decompressor = decompressor_transform

    def forward(self, x):
        x = self.encoder(x)
        if self.bottleneck_updated and not self.training:
            x = self.compressor(x)
            if self.analyzes_after_compress:
                self.analyze(x)
            x = self.decompressor(x)
        return self._tail(x)

    def update(self):
        self.bottleneck_updated = True

    def get_aux_module(self, **kwargs):
        return None


def wrap_model(wrapper_model_name, model, compression_model, **kwargs):
    if wrapper_model_name not in WRAPPER_CLASS_DICT:
        raise ValueError('wrapper_model_name `{}` is not expected'.format(wrapper_model_name))
    return WRAPPER_CLASS_DICT[wrapper_model_name](model, compression_model=compression_model, **kwargs)


def load_classification_model(model_config, device, strict=True):
    """`{key, kwargs, src_ckpt?}` -> classifier on `device` (registry.py:108-139): this build's ResNets and backbones."""
    from .backbone import get_backbone
    from .ckpt import load_ckpt
    from .resnet import RESNET_FUNC_DICT
    name = model_config['key']
    kwargs = dict(model_config.get('kwargs') or {})
    model = RESNET_FUNC_DICT[name](**kwargs) if name in RESNET_FUNC_DICT else get_backbone(name, **kwargs)
    if model is None:
        raise KeyError('classification model `{}` is not available in this build'.format(name))
    if model_config.get('src_ckpt') is not None:
        load_ckpt(model_config['src_ckpt'], model=model, strict=strict)
    return model.to(device)


def get_wrapped_classification_model(wrapper_model_config, device):
    """models.model block with a `classification_model` entry -> wrapped model (wrapper.py:343-370)."""
    from .ckpt import load_ckpt
    from .compression import get_compression_model
    name = wrapper_model_config['key']
    if name not in WRAPPER_CLASS_DICT:
        raise ValueError('wrapper_model_name `{}` is not expected'.format(name))
    compression_model = get_compression_model(wrapper_model_config.get('compression_model', None), device)
    model = load_classification_model(wrapper_model_config['classification_model'], device)
    wrapped = WRAPPER_CLASS_DICT[name](model, compression_model=compression_model, device=device,
                                       **(wrapper_model_config.get('kwargs') or {}))
    if wrapper_model_config.get('src_ckpt') is not None:
        load_ckpt(wrapper_model_config['src_ckpt'], model=wrapped, strict=False)
    return wrapped


def load_model(model_config, device):
    """script/task/image_classification.py:52-55."""
    if 'classification_model' not in model_config:
        return load_classification_model(model_config, device)
    return get_wrapped_classification_model(model_config, device)
