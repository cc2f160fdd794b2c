"""Rate loss: host-side mirror of sc2bench/loss.py:5-37 (``BppLoss``), registered as a mid-level loss.

Reads ``(features, likelihoods)`` from ``io_dict[entropy_module_path]['output']`` (the forward-hook
output of ``bottleneck_layer.entropy_bottleneck``) exactly as the reference does; note the reference
divides by the LATENT n*h*w for 'mean', never by image pixels.
"""
from torch import nn

MIDDLE_LEVEL_LOSS_DICT = dict()


def register_mid_level_loss(cls):
    MIDDLE_LEVEL_LOSS_DICT[cls.__name__] = cls
    return cls


@register_mid_level_loss
class BppLoss(nn.Module):
    """:param entropy_module_path: module path whose hooked output is (features, likelihoods)
    :param reduction: 'sum', 'batchmean' or 'mean'"""

    def __init__(self, entropy_module_path, reduction='mean'):
        super().__init__()
        self.entropy_module_path = entropy_module_path
        self.reduction = reduction

    def forward(self, student_io_dict, *args, **kwargs):
        features, likelihoods = student_io_dict[self.entropy_module_path]['output']
        n, _, h, w = features.shape
        bits = -likelihoods.log2().sum()
        if self.reduction == 'sum':
            return bits
        if self.reduction == 'batchmean':
            return bits / n
        return bits / (n * h * w)
