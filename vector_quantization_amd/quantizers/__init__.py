from .anchors import BaseAnchor, CachedAnchor, MultinomialAnchor, NearestAnchor
from .memo import ModuleDict, build_module_dict, get_memo
from .quantizer_api import BaseQuantizer
from .callbacks import (BaseCallback, ComposedCallback, CVQVAECallback, LazyInitWeightsMixin, NormalizeCallback,
                        QuantizerHolderMixin, UpdateMixin, VQKDCallback)
from .distances import BaseDistance, CosineDistance, L2Distance, LazyDistance, as_distance_tensor
from .losses import BaseLoss, CodebookLoss, CommitmentLoss, EntropyLoss, VQGANLoss
from .statistics import QuantStatistics
from .vector_quantizer import VectorQuantizer, VQGANQuantizer, VQKDQuantizer

__all__ = [
    'BaseAnchor', 'CachedAnchor', 'MultinomialAnchor', 'NearestAnchor', 'BaseQuantizer', 'ModuleDict',
    'build_module_dict', 'get_memo', 'BaseCallback', 'ComposedCallback', 'CVQVAECallback', 'LazyInitWeightsMixin',
    'NormalizeCallback', 'QuantizerHolderMixin', 'UpdateMixin', 'VQKDCallback', 'BaseDistance', 'CosineDistance',
    'L2Distance', 'LazyDistance', 'as_distance_tensor', 'BaseLoss', 'CodebookLoss', 'CommitmentLoss', 'EntropyLoss',
    'VQGANLoss', 'QuantStatistics', 'VectorQuantizer', 'VQGANQuantizer', 'VQKDQuantizer',
]
