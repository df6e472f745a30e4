"""The quantizer contract every caller of the hot path relies on.

Public surface = the reference's ``BaseQuantizer`` (vq/tasks/image_tokenization/models/quantizers/base.py:26-182): the
method names, what they return, which memo keys they fill and the order in which callbacks see the data.  The three
stages — encode, decode, loss — share one shape: callbacks may rewrite the inputs, the subclass does the work and
leaves its by-products in ``memo[stage]``, callbacks may rewrite the result.

    encode : x            -> (x', quant)        memo['encode']   hooks: before_encode(x), after_encode(x', quant)
    decode : quant        -> z                  memo['decode']   hooks: before_decode(quant), after_decode(z)
    loss   : (z, x')      -> scalar             memo['loss']     hooks: before_loss(z, x'), after_loss(loss)
    forward: encode, memo.update(x=x', quant=quant), decode, loss   ->   (z, loss, memo)
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import TYPE_CHECKING

import torch
from torch import nn

from ..config import BuildPreHookMixin, Config, Item, RegistryMeta
from ..registries import VQITQuantizerCallbackRegistry, VQITQuantizerLossRegistry
from .memo import Memo, ModuleDict, build_module_dict, get_memo

if TYPE_CHECKING:
    from .callbacks import ComposedCallback


class _CodebookShape(ABC):
    """What model builders and metrics read off a quantizer (models/base.py:83, runners/metrics.py:35)."""

    @property
    @abstractmethod
    def embedding_dim(self) -> int:
        """D"""

    @property
    @abstractmethod
    def codebook_size(self) -> int:
        """K"""

    @property
    @abstractmethod
    def embeddings(self) -> torch.Tensor:
        """a copy of the [K, D] codebook"""


class BaseQuantizer(BuildPreHookMixin, _CodebookShape, nn.Module):
    # sub-configs turned into objects before __init__ runs, in this order (each has a `<name>_build_pre_hook`)
    _BUILT_PARTS = ('callbacks', 'losses')
    # False = the reference's semantics: update callbacks rebind weight.data / re-register buffers with fresh tensors.
    # True  = they write the same values into the existing storage, which a captured HIP graph requires (graphs.py).
    inplace_updates = False

    def __init__(self, *args, callbacks: 'ComposedCallback', losses: ModuleDict, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._callbacks, self._losses = callbacks, losses
        self._init()

    def _init(self) -> None:
        self._callbacks.bind(self)

    # ---- construction from config --------------------------------------------------------------------------------

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        for part in cls._BUILT_PARTS:
            config = getattr(cls, part + '_build_pre_hook')(config, registry, item)
        return config

    @classmethod
    def callbacks_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        """A list of callback configs becomes one ComposedCallback (priorities are resolved in there)."""
        from .callbacks import ComposedCallback
        composed = Config(type=ComposedCallback.__name__, callbacks=config.get('callbacks', []))
        config.callbacks = VQITQuantizerCallbackRegistry.build(composed)
        return config

    @classmethod
    def losses_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config.losses = build_module_dict(VQITQuantizerLossRegistry, config.get_config('losses'))
        return config

    # ---- weights ---------------------------------------------------------------------------------------------------

    def _init_weights(self, config: Config) -> bool:
        return True

    def init_weights(self, config: Config) -> bool:
        """Callbacks get their own sections of the config ('before_init_weights' / 'after_init_weights'); the latter
        decides whether initialisation recurses into child modules."""
        config = Config(config)
        hook_cfg = {when: config.pop(when + '_init_weights', Config()) for when in ('before', 'after')}
        self._callbacks.before_init_weights(hook_cfg['before'])
        recursive = self._init_weights(config)
        return self._callbacks.after_init_weights(hook_cfg['after'], recursive)

    # ---- the work subclasses supply ---------------------------------------------------------------------------------

    @abstractmethod
    def _encode(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        """latents [N, D] -> (tokens int64 [N], stage memo)"""

    @abstractmethod
    def _decode(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        """tokens [*] -> (codebook rows [*, D], stage memo)"""

    def _loss(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        """Every configured loss on (z, x); the per-loss values stay in the stage memo, their fp32 sum is the result."""
        per_loss: dict[str, torch.Tensor] = self._losses(z, x, memo)
        memo.update(per_loss)
        total = x.new_zeros([], dtype=torch.float32)
        for value in per_loss.values():
            total = total + value
        return total, memo

    # ---- the three hooked stages and their composition ---------------------------------------------------------------

    def encode(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        hooks = self._callbacks
        x = hooks.before_encode(x, memo)                        # may replace x (NormalizeCallback)
        quant, memo['encode'] = self._encode(x, get_memo(memo, 'encode'))
        return x, hooks.after_encode(x, quant, memo), memo

    def decode(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        hooks = self._callbacks
        z, memo['decode'] = self._decode(hooks.before_decode(quant, memo), get_memo(memo, 'decode'))
        return hooks.after_decode(z, memo), memo

    def loss(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        hooks = self._callbacks
        z, x = hooks.before_loss(z, x, memo)
        value, memo['loss'] = self._loss(z, x, get_memo(memo, 'loss'))
        return hooks.after_loss(value, memo), memo

    def forward(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        x, quant, memo = self.encode(x, memo)
        memo.update(x=x, quant=quant)                           # what decode-side callbacks and the STE read
        z, memo = self.decode(quant, memo)
        value, memo = self.loss(z, x, memo)
        return z, value, memo
