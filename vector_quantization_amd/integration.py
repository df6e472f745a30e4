"""Registers this package's classes over the reference's registry entries (option A).

Usable where the reference (`vq`) and its framework dependency (`todd`) are importable.  It touches nothing but
`Registry.register_`, the mechanism the reference's own `custom_imports` modules use (configs/vqgan/custom_imports.py:1-3),
and is executed against the reference's REAL registry classes in the build container
(tests/test_reference_pin.py::test_register_into_reference_registries: the reference's files loaded by oracle/ref_import.py
behind a structure-only stand-in for the un-vendored todd).
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
