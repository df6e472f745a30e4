"""vector_quantization_amd — MI355X-native VQ codebook lookup (distance → argmin → gather/STE/loss →
codebook update) behind the quantizer API of magic-research/vector_quantization.

The arithmetic lives in libvqhip.so (hand-written HIP for gfx950, C ABI in include/vqhip.h);
this package is the thin Python host side: the reference's quantizer nn.Modules, registries and callbacks
re-expressed on top of that library.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .config import BuildPreHookMixin, Config, Registry, RegistryMeta
from .registries import (AnchorRegistry, InitRegistry, ModelRegistry, VQITQuantizerCallbackRegistry,
                         VQITQuantizerDistanceRegistry, VQITQuantizerLossRegistry, VQITQuantizerRegistry)
from .utils import EMA, ema

__all__ = [
    'BuildPreHookMixin', 'Config', 'Registry', 'RegistryMeta', 'AnchorRegistry', 'InitRegistry', 'ModelRegistry',
    'VQITQuantizerCallbackRegistry', 'VQITQuantizerDistanceRegistry', 'VQITQuantizerLossRegistry',
    'VQITQuantizerRegistry', 'EMA', 'ema', 'build_quantizer',
]
__version__ = '0.1.0'


def build_quantizer(config, **kwargs):
    """VQITQuantizerRegistry.build(config) with this package's classes registered (the reference's
    ``custom_imports`` step, configs/vqgan/custom_imports.py:1-3)."""
    from . import quantizers  # noqa: F401  (registers the classes)
    return VQITQuantizerRegistry.build(config, **kwargs)
