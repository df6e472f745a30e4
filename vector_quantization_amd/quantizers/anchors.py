"""CVQ-VAE anchor samplers — mirror of vq/algorithms/cvqvae/anchors.py:23-166."""
from __future__ import annotations

import random
from abc import ABC, abstractmethod
from typing import Mapping

import torch
import torch.distributed as dist
from torch import nn

from .. import ops
from ..config import Config
from ..registries import AnchorRegistry
from ..utils import Store, get_world_size, is_sync
from .memo import Memo
from .distances import LazyDistance, as_distance_tensor


def all_gather(t: torch.Tensor) -> list:
    out = [torch.empty_like(t) for _ in range(get_world_size())]
    dist.all_gather(out, t.contiguous())
    return out


class BaseAnchor(nn.Module, ABC):

    def __init__(self, *args, sync: bool = False, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._sync = sync

    @abstractmethod
    def _anchors(self, x, e, d, quant, p, memo: Memo) -> tuple[torch.Tensor, Memo]:
        pass

    def _gather(self, x, e, d, quant, p):
        """sync=True (anchors.py:50-57).  The reference all-gathers the [N, K] matrix; here only the latents travel
        and the matrix-dependent samplers rebuild what they need from the gathered rows."""
        x = torch.cat(all_gather(x))
        if Store.DRY_RUN:
            assert is_sync(e)
        quant = torch.cat(all_gather(quant))
        p = torch.stack(all_gather(p)).mean(0)
        if isinstance(d, LazyDistance):
            d = LazyDistance(d._distance, x, e)
        else:
            d = torch.cat(all_gather(d))
        return x, d, quant, p

    def forward(self, x, e, d, quant, p, memo: Memo | None = None) -> tuple[torch.Tensor, Memo]:
        if self._sync and get_world_size() > 1:
            x, d, quant, p = self._gather(x, e, d, quant, p)
        if memo is None:
            memo = Config()
        anchors, memo = self._anchors(x, e, d, quant, p, memo)
        if self._sync:
            if Store.DRY_RUN:
                assert is_sync(anchors)
        elif get_world_size() > 1:
            dist.all_reduce(anchors)
            anchors /= get_world_size()
        return anchors, memo


@AnchorRegistry.register_()
class NearestAnchor(BaseAnchor):
    """anchors[k] = x[argmin_n d[n, k]] — fused column argmin, the matrix is never formed."""

    def _anchors(self, x, e, d, quant, p, memo: Memo):
        indices = d.argmin(0)                   # LazyDistance.argmin(0) → vqhip_col_argmin; a tensor → torch
        anchors = ops.gather_rows(x, indices)
        return anchors, memo


@AnchorRegistry.register_()
class MultinomialAnchor(BaseAnchor):
    """anchors.py:88-104 — needs the materialised matrix; not used by any shipped config (fallback, unoptimised)."""

    def _anchors(self, x, e, d, quant, p, memo: Memo):
        d = as_distance_tensor(d)
        indices = d.t().softmax(1).multinomial(1).reshape(-1)
        return ops.gather_rows(x, indices), memo


@AnchorRegistry.register_()
class CachedAnchor(BaseAnchor):
    """anchors.py:107-166 — random permutation with a cache of the previous anchors (fallback, unoptimised)."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._update_cache(torch.empty(0))

    @property
    def cache(self) -> torch.Tensor:
        return self.get_buffer('_cache')

    def _update_cache(self, value: torch.Tensor) -> None:
        self.register_buffer('_cache', value)

    def _load_from_state_dict(self, state_dict: Mapping[str, torch.Tensor], prefix: str, *args, **kwargs) -> None:
        cache = state_dict.get(f'{prefix}_cache')
        if cache is not None:
            self.cache.resize_(cache.shape)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _anchors(self, x, e, d, quant, p, memo: Memo):
        K = e.shape[0]
        x = x.float()
        if x.shape[0] < K and self.cache.numel() > 0:
            x = torch.cat([x, self.cache.to(x.device)])
        indices = torch.randperm(K) if x.shape[0] <= K else torch.as_tensor(random.sample(range(x.shape[0]), K))
        if x.shape[0] < K:
            missing = torch.rand(K - x.shape[0], x.shape[1], device=x.device)
            x = torch.cat([x, missing])
        anchors = ops.gather_rows(x, indices.to(x.device))
        return anchors, memo

    def forward(self, *args, **kwargs):
        anchors, memo = super().forward(*args, **kwargs)
        self._update_cache(anchors.detach())
        return anchors, memo
