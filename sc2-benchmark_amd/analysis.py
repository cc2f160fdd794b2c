"""Data-size analysis: host-side mirror of sc2bench/analysis.py (AnalyzableModule :24-80,
FileSizeAnalyzer :110-148, FileSizeAccumulator :151-171, get_analyzer).

The reported "data size" of the reference is ``sys.getsizeof(pickle.dumps(obj)) / unit`` over the whole
``{'strings': [[bytes, ...]], 'shape': torch.Size}`` dict (torchdistill ``get_binary_object_size`` at
analysis.py:133), i.e. it includes pickle framing and CPython's bytes header; reproduced byte for byte.
"""
import logging
import pickle
import sys

import numpy as np
from torch import nn

logger = logging.getLogger(__name__)
ANALYZER_CLASS_DICT = dict()


def register_analysis_class(cls):
    ANALYZER_CLASS_DICT[cls.__name__] = cls
    return cls


def get_binary_object_size(obj, unit_size=1024):
    """torchdistill.common.file_util.get_binary_object_size."""
    return sys.getsizeof(pickle.dumps(obj)) / unit_size


class AnalyzableModule(nn.Module):
    """Module that can hand intermediate (compressed) objects to a list of analyzers."""

    def __init__(self, analyzer_configs=None):
        super().__init__()
        self.analyzers = [get_analyzer(cfg['key'], **cfg['kwargs']) for cfg in (analyzer_configs or list())]
        self.activated_analysis = False

    def forward(self, *args, **kwargs):
        raise NotImplementedError()

    def activate_analysis(self):
        self.activated_analysis = True

    def deactivate_analysis(self):
        self.activated_analysis = False

    def analyze(self, compressed_obj):
        if not self.activated_analysis:
            return
        for analyzer in self.analyzers:
            analyzer.analyze(compressed_obj)

    def summarize(self):
        for analyzer in self.analyzers:
            analyzer.summarize()

    def clear_analysis(self):
        for analyzer in self.analyzers:
            analyzer.clear()


class BaseAnalyzer(object):
    def analyze(self, *args, **kwargs):
        raise NotImplementedError()

    def summarize(self):
        raise NotImplementedError()

    def clear(self):
        raise NotImplementedError()


@register_analysis_class
class FileSizeAnalyzer(BaseAnalyzer):
    """Pickled size of each compressed object, in B / KB / MB."""
    UNIT_DICT = {'B': 1, 'KB': 1024, 'MB': 1024 * 1024}

    def __init__(self, unit='KB', **kwargs):
        self.unit = unit
        self.unit_size = self.UNIT_DICT[unit]
        self.kwargs = kwargs
        self.file_size_list = list()

    def analyze(self, compressed_obj):
        self.file_size_list.append(get_binary_object_size(compressed_obj, unit_size=self.unit_size))

    def summary(self):
        sizes = np.array(self.file_size_list)
        return {'unit': self.unit, 'mean': float(sizes.mean()), 'std': float(sizes.std()), 'count': len(sizes)}

    def summarize(self):
        s = self.summary()
        logger.info('Bottleneck size [{}]: mean {} std {} for {} samples'.format(s['unit'], s['mean'], s['std'],
                                                                                 s['count']))

    def clear(self):
        self.file_size_list.clear()


@register_analysis_class
class FileSizeAccumulator(FileSizeAnalyzer):
    """Stores pre-computed sizes (bytes) instead of measuring objects."""

    def analyze(self, file_size):
        self.file_size_list.append(file_size / self.unit_size)


def get_analyzer(cls_name, **kwargs):
    if cls_name not in ANALYZER_CLASS_DICT:
        return None
    return ANALYZER_CLASS_DICT[cls_name](**kwargs)
