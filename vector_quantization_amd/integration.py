"""Registers this package's classes over the reference's registry entries (option A).

Usable only where the reference (`vq`) and its framework dependency (`todd`) are importable — neither is in the
build image, so this module is exercised by construction only: it touches nothing but `Registry.register_`, the
mechanism the reference's own `custom_imports` modules use (configs/vqgan/custom_imports.py:1-3).
"""
from __future__ import annotations

REPLACED = {
    'VQITQuantizerRegistry': ('VectorQuantizer', 'VQGANQuantizer', 'VQKDQuantizer'),
    'VQITQuantizerDistanceRegistry': ('L2Distance', 'CosineDistance'),
    'VQITQuantizerLossRegistry': ('CodebookLoss', 'CommitmentLoss', 'VQGANLoss', 'EntropyLoss'),
    'VQITQuantizerCallbackRegistry': ('ComposedCallback', 'NormalizeCallback', 'VQKDCallback', 'CVQVAECallback'),
    'AnchorRegistry': ('NearestAnchor', 'MultinomialAnchor', 'CachedAnchor'),
    'VQITConnectorRegistry': ('BaseConnector', 'ConvConnector'),
}


def register_into_reference() -> dict:
    """Force-register the MI355X implementations under the reference's names; returns {registry: [names]}."""
    from vq.algorithms.cvqvae.registries import AnchorRegistry  # type: ignore  # noqa: I001
    from vq.algorithms.vq.distances import VQITQuantizerDistanceRegistry  # type: ignore
    from vq.tasks.image_tokenization.models.quantizers.registries import (  # type: ignore
        VQITQuantizerCallbackRegistry, VQITQuantizerLossRegistry)
    from vq.tasks.image_tokenization.models.registries import (  # type: ignore
        VQITConnectorRegistry, VQITQuantizerRegistry)

    from . import connectors as C, quantizers as Q

    registries = dict(VQITQuantizerRegistry=VQITQuantizerRegistry,
                      VQITQuantizerDistanceRegistry=VQITQuantizerDistanceRegistry,
                      VQITQuantizerLossRegistry=VQITQuantizerLossRegistry,
                      VQITQuantizerCallbackRegistry=VQITQuantizerCallbackRegistry, AnchorRegistry=AnchorRegistry,
                      VQITConnectorRegistry=VQITConnectorRegistry)
    done = {}
    for reg_name, names in REPLACED.items():
        for n in names:
            registries[reg_name].register_(n, force=True)(getattr(C if reg_name == 'VQITConnectorRegistry' else Q, n))
        done[reg_name] = list(names)
    return done
