"""Quantizer callbacks — mirror of
  vq/tasks/image_tokenization/models/quantizers/callbacks/{base,composed,lazy_init_weights}.py,
  .../quantizers/utils/quantizer_holder.py:15-27,
  vq/algorithms/vq/callbacks/{normalize,update}.py,
  vq/algorithms/vqkd/quantizers/callbacks.py:26-129 and vq/algorithms/cvqvae/quantizer_callback.py:25-105.
Hook names, order and side effects follow the reference; the arithmetic runs in libvqhip."""
from __future__ import annotations

import random
from abc import abstractmethod
from typing import TYPE_CHECKING, Iterable, Mapping

import torch
import torch.distributed as dist

from .. import ops
from ..config import BuildPreHookMixin, Config, Item, RegistryMeta
from ..registries import AnchorRegistry, VQITQuantizerCallbackRegistry
from ..utils import (EMA, PriorityQueue, Store, all_reduce_statistics, broadcast_, gather_to_rank0, get_rank,
                     get_world_size, is_sync)
from .anchors import NearestAnchor
from .distances import LazyDistance
from .memo import Memo
from .quantizer_api import BaseQuantizer
from .statistics import QuantStatistics

if TYPE_CHECKING:
    from .vector_quantizer import VectorQuantizer


class QuantizerHolderMixin:
    """todd.utils.HolderMixin[BaseQuantizer]: a non-Module holder bound to its quantizer."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__()
        self._instance = None

    def bind(self, instance: BaseQuantizer) -> None:
        self._instance = instance

    @property
    def quantizer(self) -> BaseQuantizer:
        return self._instance

    @property
    def vector_quantizer(self) -> 'VectorQuantizer':
        from .vector_quantizer import VectorQuantizer
        assert isinstance(self.quantizer, VectorQuantizer)
        return self.quantizer


class BaseCallback(QuantizerHolderMixin):

    def before_init_weights(self, config: Config) -> None:
        pass

    def after_init_weights(self, config: Config, recursive: bool) -> bool:
        return recursive

    def before_encode(self, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        return x

    def after_encode(self, x: torch.Tensor, quant: torch.Tensor, memo: Memo) -> torch.Tensor:
        return quant

    def before_decode(self, quant: torch.Tensor, memo: Memo) -> torch.Tensor:
        return quant

    def after_decode(self, z: torch.Tensor, memo: Memo) -> torch.Tensor:
        return z

    def before_loss(self, z: torch.Tensor, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor]:
        return z, x

    def after_loss(self, loss: torch.Tensor, memo: Memo) -> torch.Tensor:
        return loss


_DECODE_LOSS_HOOKS = ('before_decode', 'after_decode', 'before_loss', 'after_loss')

# The nine hooks of the protocol (callbacks/composed.py:16-19) -> positions of the arguments a callback may replace by its
# return value: none for the two notification hooks, one for the value-threading hooks, two for before_loss.
_HOOK_THREADS = {
    'bind': (), 'before_init_weights': (),
    'after_init_weights': (1,),          # (config, recursive) -> recursive
    'before_encode': (0,),               # (x, memo) -> x
    'after_encode': (1,),                # (x, quant, memo) -> quant
    'before_decode': (0,),               # (quant, memo) -> quant
    'after_decode': (0,),                # (z, memo) -> z
    'before_loss': (0, 1),               # (z, x, memo) -> (z, x)
    'after_loss': (0,),                  # (loss, memo) -> loss
}


@VQITQuantizerCallbackRegistry.register_()
class ComposedCallback(BuildPreHookMixin, BaseCallback):
    """Fans every hook out to its callbacks in per-hook priority order (ascending, stable), threading the value each
    hook may rewrite from one callback to the next (callbacks/composed.py:22-108).  The dispatchers are generated from
    ``_HOOK_THREADS`` below the class."""

    def __init__(self, *args, priorities: Iterable[Mapping[str, int]], callbacks: Iterable[BaseCallback], **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._priority_queue = PriorityQueue(priorities, callbacks)

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        entries = [Config(c) if isinstance(c, dict) else c for c in config.callbacks]
        config.priorities = [c.pop('priority', dict()) if isinstance(c, dict) else dict() for c in entries]
        config.callbacks = [registry.build_or_return(c) for c in entries]
        return config

    @property
    def callbacks(self) -> list:
        return self._priority_queue('bind')

    def overrides_decode_or_loss(self) -> bool:
        """True when some callback customises a decode/loss hook (the fused decode+loss path must then be skipped)."""
        for cb in self.callbacks:
            for name in _DECODE_LOSS_HOOKS:
                if getattr(type(cb), name) is not getattr(BaseCallback, name):
                    return True
        return False


def _make_dispatcher(hook: str, threaded: tuple):
    own = getattr(BaseCallback, hook)                       # the composed callback's own (no-op / bind) behaviour runs first

    def dispatch(self, *args, **kwargs):
        args = list(args)
        for target in [lambda *a, **k: own(self, *a, **k)] + [getattr(cb, hook) for cb in self._priority_queue(hook)]:
            out = target(*args, **kwargs)
            if len(threaded) == 1:
                args[threaded[0]] = out
            elif threaded:
                for pos, value in zip(threaded, out):
                    args[pos] = value
        if not threaded:
            return None
        return args[threaded[0]] if len(threaded) == 1 else tuple(args[pos] for pos in threaded)

    dispatch.__name__ = dispatch.__qualname__ = hook
    dispatch.__doc__ = f'{hook}: own behaviour, then every callback in priority order' + \
        (f' (argument(s) {threaded} threaded through the returns)' if threaded else '')
    return dispatch


for _hook, _threaded in _HOOK_THREADS.items():
    setattr(ComposedCallback, _hook, _make_dispatcher(_hook, _threaded))


class LazyInitWeightsMixin(BaseCallback):

    @abstractmethod
    def lazy_init_weights(self, config: Config, x: torch.Tensor, memo: Memo) -> None:
        pass

    def before_init_weights(self, config: Config) -> None:
        super().before_init_weights(config)
        lazy_init_weights = config.pop('lazy_init_weights', Config())

        def forward_pre_hook(module: BaseQuantizer, args: tuple[torch.Tensor, Memo]) -> None:
            x, memo = args
            self.lazy_init_weights(lazy_init_weights, x, memo)
            handle.remove()

        handle = self.quantizer.register_forward_pre_hook(forward_pre_hook)


class UpdateMixin(BuildPreHookMixin, BaseCallback):

    def __init__(self, *args, ema: EMA | None = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if ema is not None:
            self._ema = ema

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        if (ema := config.get('ema')) is not None:
            config.ema = EMA(**ema)
        return config

    @property
    def with_ema(self) -> bool:
        return hasattr(self, '_ema')

    def _update_embedding(self, e: torch.Tensor) -> None:
        if Store.DRY_RUN:
            assert is_sync(e)
        weight = self.vector_quantizer.embedding.weight
        if self.quantizer.inplace_updates:                   # same values into the existing storage (graph replay)
            weight.data.copy_(e)
        else:
            weight.data = e                                  # rebinds the storage, like callbacks/update.py:56
        self.vector_quantizer.invalidate_codebook()          # neither form bumps weight._version


@VQITQuantizerCallbackRegistry.register_()
class NormalizeCallback(UpdateMixin, BaseCallback):

    def before_encode(self, x: torch.Tensor, memo: Memo) -> torch.Tensor:
        x = super().before_encode(x, memo)
        from .. import functional as VF
        x = VF.normalize(x)                                   # F.normalize(x): differentiable w.r.t. the encoder
        e = self.vector_quantizer.embedding.weight
        e = ops.normalize_rows(e.detach())
        self._update_embedding(e)
        return x


def distributed_cat(x: torch.Tensor) -> torch.Tensor:
    """The first batch of every rank, concatenated on rank 0; an empty tensor on the other ranks
    (vqkd/quantizers/callbacks.py:26-35)."""
    if get_world_size() <= 1:
        return x
    parts = gather_to_rank0(x)
    return x.new_empty(0) if parts is None else torch.cat(parts)


@VQITQuantizerCallbackRegistry.register_()
class VQKDCallback(LazyInitWeightsMixin, NormalizeCallback):

    def _statistics(self, x: torch.Tensor, quant: torch.Tensor, sync: bool):
        """Histogram + per-code sums of the assigned (already normalised) latents, all-reduced when syncing."""
        K = self.vector_quantizer.codebook_size
        hist = ops.hist(quant, K).to(torch.int64)
        sums = ops.scatter_add_rows(x, quant, K)
        if sync and get_world_size() > 1:
            hist, _, sums = all_reduce_statistics(hist, quant.numel(), sums)
        return hist, sums

    def _kmeans(self, x: torch.Tensor, quant: torch.Tensor, sync: bool) -> torch.Tensor:
        """callbacks.py:44-71 — centroids, old row kept where a code received no token."""
        e = self.vector_quantizer.embeddings
        hist, sums = self._statistics(x, quant, sync)
        ops.vqkd_update_(e, hist, sums, 0.0, mode='centroid')
        return e

    def _update_embedding(self, e: torch.Tensor) -> None:
        e = ops.normalize_rows(e)
        return super()._update_embedding(e)

    def lazy_init_weights(self, config: Config, x: torch.Tensor, memo: Memo) -> None:
        if not self.quantizer.training:
            return
        x = distributed_cat(x.detach())
        e = self.vector_quantizer.embeddings
        iters = config.get('iters', 10)
        if get_rank() > 0:
            e = torch.empty_like(e)
        elif x.shape[0] < e.shape[0]:
            e[:x.shape[0]] = x
        else:
            x = ops.normalize_rows(x)
            # (the reference offloads to the CPU when N*K > 2^30 because it materialises d[N, K]; the fused
            #  argmin never forms the matrix, so the Lloyd iterations stay on the device)
            indices = random.sample(range(x.shape[0]), e.shape[0])
            e = ops.gather_rows(x, torch.as_tensor(indices, device=x.device))
            for _ in range(iters):
                self._update_embedding(e)
                quant, _ = self.vector_quantizer._encode(x, Config())
                e = self._kmeans(x, quant, False)
        if get_world_size() > 1:
            broadcast_(e, 0)
        self._update_embedding(e)

    def after_encode(self, x: torch.Tensor, quant: torch.Tensor, memo: Memo) -> torch.Tensor:
        quant = super().after_encode(x, quant, memo)
        if not self.quantizer.training:
            return quant
        x = ops.normalize_rows(x.detach())                     # callbacks.py:124
        hist, sums = self._statistics(x, quant, True)          # :125 (hist + sums, two collectives at most)
        e = self.vector_quantizer.embeddings                   # clone: the update writes a fresh tensor
        ops.vqkd_update_(e, hist, sums, self._ema.decay)       # :66-70,126-127 and the final normalise (:73-75)
        UpdateMixin._update_embedding(self, e)
        return quant


@VQITQuantizerCallbackRegistry.register_()
class CVQVAECallback(UpdateMixin, BaseCallback):

    def __init__(self, *args, anchor, eps: float = 1e-3, sparse_anchors: bool = False, **kwargs) -> None:
        """``sparse_anchors`` (extension, default off = the reference's data flow): anchors are computed, all-reduced
        and applied only for the codes whose decay is below 1 — every code in regular use has decay == 1.0f exactly
        and its anchor is multiplied by 0.  Same result (see include/vqhip.h, vqhip_cvq_update_rows), a column argmin
        over a fraction of the codebook and an [M, D] instead of a [K, D] all-reduce; costs one host synchronisation
        per step (the number of such codes sizes the exchange).  NearestAnchor only."""
        super().__init__(*args, **kwargs)
        self._anchor = anchor
        self._eps = eps
        self._sparse_anchors = sparse_anchors

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        config.anchor = AnchorRegistry.build_or_return(config.anchor)
        return config

    def before_init_weights(self, config: Config) -> None:
        super().before_init_weights(config)
        if not self.quantizer.training:
            return
        p = torch.zeros(self.quantizer.codebook_size, device=self.vector_quantizer.embedding.weight.device)
        self._update_probability(p)

    @property
    def probability(self) -> torch.Tensor:
        return self.quantizer.get_buffer('_probability')

    def _update_probability(self, value: torch.Tensor) -> None:
        if self.quantizer.inplace_updates and '_probability' in self.quantizer._buffers \
                and self.quantizer._buffers['_probability'].shape == value.shape \
                and self.quantizer._buffers['_probability'].device == value.device:
            self.quantizer._buffers['_probability'].copy_(value)
        else:
            self.quantizer.register_buffer('_probability', value)

    def after_encode(self, x: torch.Tensor, quant: torch.Tensor, memo: Memo) -> torch.Tensor:
        quant = super().after_encode(x, quant, memo)
        if not self.quantizer.training:
            return quant
        K = self.quantizer.codebook_size
        d = memo['encode']['distance']
        hist32 = memo['encode'].get('hist')
        if (get_world_size() <= 1 and type(self._anchor) is NearestAnchor and not self._anchor._sync
                and not self._sparse_anchors and isinstance(d, LazyDistance) and hist32 is not None
                and self.probability.is_cuda and self.probability.dtype == torch.float32):
            # one rank: nothing is exchanged between the probability update and the blend, so the whole update is one
            # launch on the epilogue histogram, the column argmin and the latents (vqhip_cvq_step; bit-identical to the
            # staged form below, which stays for the multi-rank case where hist and anchors are all-reduced in between)
            weight = self.vector_quantizer.embedding.weight
            col = d.argmin(0)
            w_in, p_in = weight.detach(), self.probability
            inplace = self.quantizer.inplace_updates
            w_out = w_in if inplace else torch.empty_like(w_in)
            p_out = p_in if inplace else torch.empty_like(p_in)
            ops.cvq_step(w_in.contiguous(), w_out, p_in.contiguous(), p_out, hist32, quant.numel(), x.detach(), col,
                         self._ema.decay, self._eps)
            if not inplace:
                self._update_probability(p_out)
                self._update_embedding(w_out)
            else:
                if Store.DRY_RUN:
                    assert is_sync(w_out)
                self.vector_quantizer.invalidate_codebook()
            return quant
        e = self.quantizer.embeddings
        stats = QuantStatistics(quant=quant, codebook_size=K, sync=True, hist=memo['encode'].get('hist'))
        hist, numel = stats.bin_count(), stats._statistics()[1]
        p = self.probability.to(device=e.device, dtype=torch.float32).clone()
        ops.cvq_update_(e, p, hist, numel, None, self._ema.decay, self._eps, stage=1)      # p = ema(p, hist/numel)
        self._update_probability(p)
        if self._sparse_anchors and isinstance(self._anchor, NearestAnchor) and isinstance(d, LazyDistance):
            decay = ops.cvq_decay(p, K, self._ema.decay, self._eps)
            rows = torch.nonzero(decay < 1.0).reshape(-1)      # host sync; the same set on every rank (p is synchronised)
            if rows.numel():
                e_sub = e.index_select(0, rows)
                d_sub = LazyDistance(d._distance, x.detach(), e_sub)
                anchors, memo = self._anchor(x.detach(), e_sub, d_sub, quant, p.index_select(0, rows), memo=memo)
                ops.cvq_update_rows_(e, p, rows, anchors, self._ema.decay, self._eps)
            self._update_embedding(e)
            return quant
        anchors, memo = self._anchor(x.detach(), e, d, quant, p, memo=memo)
        # decay = 1 - exp(-p*K*10/(1-ema.decay) - eps); e = e*decay + anchors*(1-decay)
        ops.cvq_update_(e, p, None, None, anchors, self._ema.decay, self._eps, stage=2)
        self._update_embedding(e)
        return quant
