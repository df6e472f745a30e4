"""Abstract quantizer API — mirror of vq/tasks/image_tokenization/models/quantizers/base.py:26-182 (same method
names, argument meaning, memo side effects and hook order)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import TYPE_CHECKING

import torch
from torch import nn

from ..config import BuildPreHookMixin, Config, Item, RegistryMeta
from ..registries import VQITQuantizerCallbackRegistry, VQITQuantizerLossRegistry

if TYPE_CHECKING:
    from .callbacks import ComposedCallback

Memo = dict


def get_memo(memo: Memo, key: str) -> Memo:
    """vq/utils/misc.py:30-38: the sub-memo under ``key`` (created empty if absent)."""
    if key in memo:
        sub = memo[key]
        assert isinstance(sub, dict)
        return sub
    return Config()


class ModuleDict(nn.ModuleDict):
    """todd.patches.torch.ModuleDict: calling it calls every member and returns {name: output}."""

    def forward(self, *args, **kwargs) -> dict:
        return {k: m(*args, **kwargs) for k, m in self.items()}


def build_module_dict(registry: RegistryMeta, config: Config, **kwargs) -> ModuleDict:
    """vq/utils/builders.py:24-34."""
    return ModuleDict({k: registry.build_or_return(v, **kwargs) for k, v in config.items() if v is not None})


class BaseQuantizer(BuildPreHookMixin, nn.Module, ABC):

    def __init__(self, *args, callbacks: 'ComposedCallback', losses: ModuleDict, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._callbacks = callbacks
        self._losses = losses
        self._init()

    @classmethod
    def callbacks_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        from .callbacks import ComposedCallback
        config.callbacks = VQITQuantizerCallbackRegistry.build(
            Config(type=ComposedCallback.__name__, callbacks=config.get('callbacks', [])),
        )
        return config

    @classmethod
    def losses_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config.losses = build_module_dict(VQITQuantizerLossRegistry, config.get_config('losses'))
        return config

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        config = cls.callbacks_build_pre_hook(config, registry, item)
        config = cls.losses_build_pre_hook(config, registry, item)
        return config

    def _init(self) -> None:
        self._callbacks.bind(self)

    @property
    @abstractmethod
    def embedding_dim(self) -> int:
        pass

    @property
    @abstractmethod
    def codebook_size(self) -> int:
        pass

    @property
    @abstractmethod
    def embeddings(self) -> torch.Tensor:
        pass

    def _init_weights(self, config: Config) -> bool:
        return True

    def init_weights(self, config: Config) -> bool:
        config = Config(config)
        before_init_weights = config.pop('before_init_weights', Config())
        after_init_weights = config.pop('after_init_weights', Config())
        self._callbacks.before_init_weights(before_init_weights)
        recursive = self._init_weights(config)
        recursive = self._callbacks.after_init_weights(after_init_weights, recursive)
        return recursive

    @abstractmethod
    def _encode(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        pass

    def encode(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        x = self._callbacks.before_encode(x, memo)
        quant, memo['encode'] = self._encode(x, get_memo(memo, 'encode'))
        quant = self._callbacks.after_encode(x, quant, memo)
        return x, quant, memo

    @abstractmethod
    def _decode(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        pass

    def decode(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        quant = self._callbacks.before_decode(quant, memo)
        z, memo['decode'] = self._decode(quant, get_memo(memo, 'decode'))
        z = self._callbacks.after_decode(z, memo)
        return z, memo

    def _loss(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        losses: dict[str, torch.Tensor] = self._losses(z, x, memo)
        memo.update(losses)
        loss = sum(losses.values(), x.new_zeros([], dtype=torch.float32))
        return loss, memo

    def loss(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        z, x = self._callbacks.before_loss(z, x, memo)
        loss, memo['loss'] = self._loss(z, x, get_memo(memo, 'loss'))
        loss = self._callbacks.after_loss(loss, memo)
        return loss, memo

    def forward(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        x, quant, memo = self.encode(x, memo)
        memo.update(x=x, quant=quant)
        z, memo = self.decode(quant, memo)
        loss, memo = self.loss(z, x, memo)
        return z, loss, memo
