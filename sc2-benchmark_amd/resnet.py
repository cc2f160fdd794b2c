"""ResNet-50 task head (layer2..fc) with torchvision's module and state-dict names.

torchvision is not installed in this image, and the reference builds its splittable model from
``torchvision.models.resnet50`` (sc2bench/models/backbone.py:690-693), keeping ``layer2, layer3, layer4,
avgpool, fc`` (backbone.py:217-223).  This file re-declares that architecture so torchvision checkpoints
(`layer2.0.conv1.weight`, ...) load by name.  The head runs on PyTorch-ROCm ops (MIOpen), bf16
channels_last at inference: it is the caller of the bottleneck path, not part of the hand-written path
(SURVEY.md 2.2 row N9).
"""
import logging
import os
import warnings

import torch
from torch import nn

logger = logging.getLogger(__name__)


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (torchvision.ops.misc.FrozenBatchNorm2d)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer('weight', torch.ones(num_features))
        self.register_buffer('bias', torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def forward(self, x):
        scale = (self.weight * (self.running_var + self.eps).rsqrt()).reshape(1, -1, 1, 1)
        bias = self.bias.reshape(1, -1, 1, 1) - self.running_mean.reshape(1, -1, 1, 1) * scale
        return x * scale.to(x.dtype) + bias.to(x.dtype)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                               bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.training and x.is_cuda and x.dtype == torch.bfloat16:
            out = self._forward_train_hip(x)
            if out is not None:
                return out
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)

    def _forward_train_hip(self, x):
        """A block that TRAINS (stage 2, bf16 autocast, channels_last): the convs stay where they are; each norm layer with its ReLU --
        and, for the third, the residual add -- is two passes of bn.hip forward and two backward (autograd._BnActFn).  None: not
        applicable (decided before anything runs; the torch modules run then)."""
        from . import autograd as A
        ds = self.downsample
        ds_ok = ds is None or (isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d))
        if not (ds_ok and all(A.bn_module_ok(b) for b in (self.bn1, self.bn2, self.bn3) + ((ds[1],) if ds is not None else ()))):
            return None
        def conv(m, t):     # (the library's conv kernels under autograd where the layer allows, else the module)
            return A.conv_train(m, t) if A.conv_module_ok(m) else m(t)
        out = A.bn_act(self.bn1, conv(self.conv1, x), relu=True)
        out = A.bn_act(self.bn2, conv(self.conv2, out), relu=True)
        identity = x if ds is None else A.bn_act(ds[1], conv(ds[0], x), relu=False)
        return A.bn_act(self.bn3, conv(self.conv3, out), relu=True, residual=identity)


class ResNet(nn.Module):
    """torchvision-layout ResNet (Bottleneck blocks).  ``layers=(3, 4, 6, 3)`` is ResNet-50."""

    def __init__(self, layers=(3, 4, 6, 3), num_classes=1000, replace_stride_with_dilation=None, norm_layer=None,
                 zero_init_residual=False):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * Bottleneck.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck) and isinstance(m.bn3, nn.BatchNorm2d):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                norm_layer(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def _build(name, blocks, weights, kwargs):
    """`weights` / `pretrained` name torchvision's downloadable ImageNet weights.  There is no network here: the
    weights are taken from $SC2_PRETRAINED_DIR/<name>.pth (a torchvision state dict) when that file exists; otherwise
    the model keeps its random initialisation and says so with a warning (SC2_STRICT_WEIGHTS=1 turns it into an error)."""
    pretrained = kwargs.pop('pretrained', None)
    model = ResNet(blocks, **kwargs)
    if weights is not None or pretrained:
        root = os.environ.get('SC2_PRETRAINED_DIR')
        path = os.path.join(root, name + '.pth') if root else None
        if path and os.path.isfile(path):
            from .ckpt import load_ckpt
            load_ckpt(path, model=model, strict=True)
        else:
            msg = ('{}(weights={!r}): pretrained weights requested but no local file found ({}); the model is RANDOMLY '
                   'INITIALISED. Put a torchvision state dict at $SC2_PRETRAINED_DIR/{}.pth'
                   .format(name, weights if weights is not None else 'pretrained', path or 'SC2_PRETRAINED_DIR unset', name))
            if os.environ.get('SC2_STRICT_WEIGHTS') == '1':
                raise FileNotFoundError(msg)
            warnings.warn(msg)
            logger.warning(msg)
    return model


def resnet50(weights=None, progress=True, **kwargs):
    return _build('resnet50', (3, 4, 6, 3), weights, kwargs)


def resnet101(weights=None, progress=True, **kwargs):
    return _build('resnet101', (3, 4, 23, 3), weights, kwargs)


def resnet152(weights=None, progress=True, **kwargs):
    return _build('resnet152', (3, 8, 36, 3), weights, kwargs)


RESNET_FUNC_DICT = {'resnet50': resnet50, 'resnet101': resnet101, 'resnet152': resnet152}
