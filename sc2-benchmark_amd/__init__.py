"""sc2bench_amd: MI355X-native implementation of sc2bench's supervised-compression bottleneck path.

The directory is named ``sc2-benchmark_amd`` (not an importable identifier); ``sc2bench_amd.py`` at the
repository root loads it under the module name ``sc2bench_amd``.
"""
from . import hip  # noqa: F401
from .analysis import ANALYZER_CLASS_DICT, AnalyzableModule, FileSizeAccumulator, FileSizeAnalyzer  # noqa: F401
from .backbone import (BACKBONE_CLASS_DICT, BACKBONE_FUNC_DICT, MODEL_DICT, FeatureExtractionBackbone,  # noqa: F401
                       SplittableResNet, UpdatableBackbone, check_if_updatable, get_backbone, splittable_resnet)
from .entropy import (CompressionModel, EntropyBottleneck, GDN1, GaussianConditional, HipConv2d,  # noqa: F401
                      HipConvTranspose2d, LowerBound, NonNegativeParametrizer, get_scale_table)
from .layer import (LAYER_CLASS_DICT, LAYER_FUNC_DICT, BaseBottleneck, EntropyBottleneckLayer,  # noqa: F401
                    FPBasedResNetBottleneck, MSHPBasedResNetBottleneck, SHPBasedResNetBottleneck, get_layer,
                    register_layer_class, register_layer_func)
from .loss import BppLoss  # noqa: F401
from .compression import (COMPRESSION_MODEL_CLASS_DICT, COMPRESSION_MODEL_FUNC_DICT, FactorizedPrior,  # noqa: F401
                          bmshj2018_factorized, get_compression_model)
from .entropy import GDN  # noqa: F401
from .transforms import AdaptivePad, PILImageModule, PILTensorModule  # noqa: F401
from .wrapper import (WRAPPER_CLASS_DICT, CodecFeatureCompressionClassifier, CodecInputCompressionClassifier,  # noqa: F401
                      EntropicClassifier, NeuralInputCompressionClassifier, SplitClassifier, wrap_model)

from .dense import (DETECTION_MODEL_FUNC_DICT, SEGMENTATION_MODEL_FUNC_DICT, BaseRCNN, BaseSegmentationModel,  # noqa: F401
                    SegEvaluator, UpdatableBackboneWithFPN, backbone_with_fpn, deeplabv3_model, faster_rcnn_model)

from .pipeline import StagePipeline, supports_stages  # noqa: F401
from . import graphs  # noqa: F401

__version__ = '0.3.0'
