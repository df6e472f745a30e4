"""vector_quantization_amd — MI355X-native VQ codebook lookup (distance → argmin → gather/STE/loss →
codebook update) behind the quantizer API of magic-research/vector_quantization.

The arithmetic lives in libvqhip.so (hand-written HIP for gfx950, C ABI in include/vqhip.h);
this package is the thin Python host side.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
__version__ = '0.1.0'
