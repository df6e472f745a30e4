"""Losses — mirror of vq/tasks/image_tokenization/models/quantizers/losses.py:13-22 and
vq/algorithms/vq/losses.py:28-153.  State-dict layout (``_weight._steps`` buffers) follows todd's BaseLoss as pinned
by tools/convert_checkpoints.py:239-243,321-322 so that converted checkpoints load strictly."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch
from torch import nn

from .. import functional as VF
from ..config import BuildPreHookMixin, Config, Item, RegistryMeta
from ..registries import VQITQuantizerLossRegistry
from .memo import Memo
from .distances import as_distance_tensor


class _Weight(nn.Module):
    """todd's loss-weight scheduler reduced to what the path needs: a constant factor and the ``_steps`` buffer."""

    def __init__(self, value: float = 1.0) -> None:
        super().__init__()
        self._value = float(value)
        self.register_buffer('_steps', torch.tensor(1))

    def forward(self, loss: torch.Tensor) -> torch.Tensor:
        return loss if self._value == 1.0 else loss * self._value


class ToddBaseLoss(nn.Module):
    """todd.models.losses.BaseLoss: mean reduction, scalar weight."""

    def __init__(self, *args, weight: float = 1.0, **kwargs) -> None:
        super().__init__()
        self._weight = _Weight(weight)

    def _reduce_weight(self, loss: torch.Tensor) -> torch.Tensor:
        return self._weight(loss)


class ToddMSELoss(ToddBaseLoss):
    """todd.models.losses.MSELoss as fixed in SURVEY.md §8c: optional F.normalize(dim=1) of both arguments, then the
    mean-reduced squared error."""

    def __init__(self, *args, norm: bool = False, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._norm = norm

    @property
    def norm(self) -> bool:
        return self._norm

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if self._norm:
            pred, target = VF.normalize(pred), VF.normalize(target)
        return self._reduce_weight(VF.mse(pred, target))


class BaseLoss(ToddBaseLoss, ABC):

    @abstractmethod
    def forward(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        pass


class MSELoss(BuildPreHookMixin, BaseLoss, ABC):

    def __init__(self, *args, mse: ToddMSELoss, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._mse = mse

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        config.mse = ToddMSELoss(**config.get_config('mse'))
        return config

    @property
    def plain(self) -> bool:
        """True when the term is an un-normalised, unit-weight MSE (eligible for the fused forward)."""
        return (not self._mse.norm) and self._mse._weight._value == 1.0 and self._weight._value == 1.0


@VQITQuantizerLossRegistry.register_()
class CodebookLoss(MSELoss):

    def forward(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        return self._reduce_weight(self._mse(z, x.detach()))


@VQITQuantizerLossRegistry.register_()
class CommitmentLoss(MSELoss):

    def forward(self, z: torch.Tensor, x: torch.Tensor, memo: Memo | None = None) -> torch.Tensor:
        return self._reduce_weight(self._mse(z.detach(), x))


@VQITQuantizerLossRegistry.register_()
class VQGANLoss(BuildPreHookMixin, BaseLoss):

    def __init__(self, *args, codebook: CodebookLoss, commitment: CommitmentLoss, beta: float = 0.25, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._codebook = codebook
        self._commitment = commitment
        self._beta = beta

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        config.codebook = VQITQuantizerLossRegistry.build(config.get_config('codebook'), type=CodebookLoss.__name__)
        config.commitment = VQITQuantizerLossRegistry.build(config.get_config('commitment'),
                                                            type=CommitmentLoss.__name__)
        return config

    @property
    def beta(self) -> float:
        return self._beta

    @property
    def plain(self) -> bool:
        return self._codebook.plain and self._commitment.plain and self._weight._value == 1.0

    def forward(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        codebook = self._codebook(z, x, memo)
        commitment = self._commitment(z, x, memo)
        return self._reduce_weight(codebook + self._beta * commitment)


@VQITQuantizerLossRegistry.register_()
class EntropyLoss(BaseLoss):
    """vq/algorithms/vq/losses.py:130-153 ("TODO: refactor" in the reference; used by no shipped config).
    Needs the whole [N, K] matrix with autograd, which the fused path never forms: ``memo['distance']`` (a
    ``LazyDistance``) is materialised on demand by the HIP distance kernel with gradients to the latents and the
    codebook (``distances._L2Matrix`` / ``_DotMatrix``); the softmax/entropy arithmetic on it is stock device ops.
    As in the reference the loss reads ``memo['distance']`` of the memo it is handed (the encode-stage memo)."""

    def __init__(self, *args, temperature: float, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._temperature = temperature

    def forward(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        affinity = as_distance_tensor(memo['distance'])
        flat_affinity = affinity.reshape(-1, affinity.shape[-1]) / self._temperature
        probs = flat_affinity.softmax(-1)
        log_probs = torch.log_softmax(flat_affinity + 1e-5, -1)
        avg_probs = probs.mean(0)
        avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + 1e-5))
        sample_entropy = -torch.mean(torch.sum(probs * log_probs, -1))
        return self._reduce_weight(sample_entropy - avg_entropy)
