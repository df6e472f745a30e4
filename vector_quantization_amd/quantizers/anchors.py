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
from ..utils import Store, all_gather, all_reduce_sum, get_world_size, is_sync
from .memo import Memo
from .distances import LazyDistance, as_distance_tensor


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
            all_reduce_sum(anchors)
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
    """One latent per code drawn from softmax over the code's column of distances (anchors.py:88-104).  Needs the
    materialised matrix (HIP distance kernel); the draw uses the device generator.  No shipped config uses it."""

    @staticmethod
    def probabilities(d) -> torch.Tensor:
        return as_distance_tensor(d).detach().t().softmax(1)             # [K, N]

    def _anchors(self, x, e, d, quant, p, memo: Memo):
        indices = self.probabilities(d).multinomial(1).reshape(-1)
        return ops.gather_rows(x, indices), memo


@AnchorRegistry.register_()
class CachedAnchor(BaseAnchor):
    """K anchors drawn without replacement from a pool of candidate rows: this step's latents, topped up — when the
    batch is smaller than the codebook — with the anchors of the previous step (the ``_cache`` buffer) and, if that is
    still not enough, with uniform noise (vq/algorithms/cvqvae/anchors.py:107-166; no shipped config uses it).
    The draw itself uses the host-side generators the reference uses (``random.sample`` when the pool is larger than
    the codebook, the CPU ``torch.randperm`` otherwise), so a seeded run selects the same rows as the reference on any
    device; only the noise rows come from the device generator."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.register_buffer('_cache', torch.empty(0))

    @property
    def cache(self) -> torch.Tensor:
        return self.get_buffer('_cache')

    def _load_from_state_dict(self, state_dict: Mapping[str, torch.Tensor], prefix: str, *args, **kwargs) -> None:
        saved = state_dict.get(prefix + '_cache')
        if saved is not None:
            self.cache.resize_(saved.shape)          # the buffer starts empty: give it the checkpoint's shape first
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _pool(self, x: torch.Tensor, K: int) -> torch.Tensor:
        pool = x.float()
        if pool.shape[0] < K and self.cache.numel() > 0:
            pool = torch.cat([pool, self.cache.to(pool.device)])
        return pool

    @staticmethod
    def _draw(rows: int, K: int) -> torch.Tensor:
        if rows > K:
            return torch.as_tensor(random.sample(range(rows), K))
        return torch.randperm(K)                      # indices >= rows select the noise rows appended below

    def _anchors(self, x, e, d, quant, p, memo: Memo):
        K = e.shape[0]
        pool = self._pool(x, K)
        picks = self._draw(pool.shape[0], K)
        short = K - pool.shape[0]
        if short > 0:
            pool = torch.cat([pool, torch.rand(short, pool.shape[1], device=pool.device)])
        return ops.gather_rows(pool, picks.to(pool.device)), memo

    def forward(self, *args, **kwargs):
        anchors, memo = super().forward(*args, **kwargs)
        self.register_buffer('_cache', anchors.detach())
        return anchors, memo
