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

from .. import exchange, ops
from ..config import BuildPreHookMixin, Config, Item, RegistryMeta
from ..registries import AnchorRegistry, VQITQuantizerCallbackRegistry
from ..utils import (EMA, PriorityQueue, Store, all_reduce_statistics, broadcast_, exchange_log, exchanging, gather_to_rank0,
                     get_rank, get_world_size, is_sync)
from .anchors import NearestAnchor
from .distances import LazyDistance
from .memo import Memo, get_memo
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

    def _writes_in_place(self, weight: torch.Tensor) -> bool:
        """True when the codebook update must land in the parameter's existing storage: graph replay (``inplace_updates``),
        or a parameter that FSDP manages (``use_orig_params=True``, configs/strategies/fsdp.py:5-8: the parameter is a view
        into FSDP's flat parameter, and a rebound ``.data`` would be dropped at the next unshard — measured: the CVQ-VAE update
        of every step was lost, tools/debug/fsdp_diff.py)."""
        # (inside an FSDP forward the module attribute is not even the Parameter but a temporary view of the flat parameter —
        #  a non-leaf tensor: rebinding ITS .data, as callbacks/update.py:56 does, changes nothing that outlives the forward)
        return bool(self.quantizer.inplace_updates or getattr(weight, '_fsdp_flattened', False) or not weight.is_leaf or weight._is_view())

    def _update_embedding(self, e: torch.Tensor) -> None:
        if Store.DRY_RUN:
            assert is_sync(e)
        weight = self.vector_quantizer.embedding.weight
        if self._writes_in_place(weight):                    # same values into the existing storage (graph replay, FSDP)
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
        hist = ops.hist(quant, K)
        sums = ops.scatter_add_rows(x, quant, K)
        if sync and exchanging():
            if get_world_size() > exchange.MAX_WORLD:      # beyond the exact range of the fp32 count pieces: the reference's
                hist, _, sums = all_reduce_statistics(hist, quant.numel(), sums)      # unpacked flow (int64 counts, two collectives)
                return hist, sums
            hist, _, sums = exchange.all_reduce_packed(hist, quant.numel(), sums)     # histogram, token count, K x D sums: ONE collective
            return hist, sums
        return hist.to(torch.int64), sums

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

    # ---- the whole training forward as ONE library call (train_step.py, include/vqhip.h: vqhip_vqkd_forward) ------------
    def fused_forward_ok(self, x: torch.Tensor) -> bool:
        """True when this step can be enqueued by one call: train mode, the lazy init done, device latents with a proposal
        image, cosine distance with the library's fused encode, an EMA, a world the packed exchange covers."""
        from .distances import CosineDistance
        q = self.vector_quantizer
        if not (q.training and self.with_ema and x.dim() == 2 and x.is_cuda and x.shape[0] > 0 and x.shape[0] < (1 << 31)):
            return False
        if type(q.distance) is not CosineDistance or not ops.coarse_supported(q.embedding_dim) or q._cache_codebook:
            return False
        if len(q._forward_pre_hooks) > 0 or get_world_size() > exchange.MAX_WORLD:
            return False
        w = q.embedding.weight
        return w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()

    def fused_forward(self, x: torch.Tensor, memo: Memo):
        """NormalizeCallback.before_encode + _encode + after_encode + decode + CommitmentLoss(norm=True) + STE of one training
        step (normalize.py:22-29, vqkd callbacks.py:114-129, quantizers.py:92-117) from one host call.  Returns
        (xn, quant, z_ste, loss) with the reference's memo side effects; weight.data is rebound (or overwritten in place)
        exactly as the two ``_update_embedding`` calls of the unfused flow leave it."""
        from .. import functional as VF, train_step
        from ..utils import all_reduce_sum
        q = self.vector_quantizer
        weight = q.embedding.weight
        K, D = weight.shape
        if getattr(self, '_step_state', None) is None:
            self._step_state = train_step.VqkdStepState()
        inplace = self._writes_in_place(weight)
        w_in = weight.detach()
        w_mid = w_in if inplace else torch.empty_like(w_in)
        w_out = w_in if inplace else torch.empty_like(w_in)
        xd = x.detach()
        exch = exchanging()
        comm = None
        if exch and not exchange_log.enabled:
            from .. import rccl
            comm = rccl.communicator(w_in)
        metric = q.distance.metric_for(D)
        ordered = ops.use_ordered(K, D, None, xd.shape[0])
        out = train_step.vqkd_forward(xd, w_in, w_mid, w_out, metric, self._ema.decay, self._step_state, exchange=exch,
                                      world=get_world_size(), comm=comm, all_reduce=all_reduce_sum if exch else None,
                                      ordered=ordered, tail=True)
        if Store.DRY_RUN:
            assert is_sync(w_out)
        if not inplace:
            weight.data = w_out                                  # callbacks/update.py:56 (after the EMA update: callbacks.py:128)
        q.invalidate_codebook()
        done = VF._Computed(xn=out['xn'], z_ste=out['z_ste'], mse=out['mse'], idx=out['idx'])
        xn, z_ste, loss = VF.vqkd_step(x, weight, done)
        enc = get_memo(memo, 'encode')
        prepared = out['prepared']
        enc['distance'] = LazyDistance(q.distance, xn, w_mid, xq=out['xq'], eq=prepared.exact_rows(), metric=ops.metric_name(prepared.metric))
        enc['hist'] = out['hist']
        memo['encode'] = enc
        return xn, out['idx'], z_ste, loss

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

    def __init__(self, *args, anchor, eps: float = 1e-3, sparse_anchors: bool | None = None, **kwargs) -> None:
        """``sparse_anchors`` (extension; None = automatic, the default): anchors are computed, exchanged and applied only
        for the codes whose decay can come out below 1 — every code in regular use has decay == 1.0f exactly and its anchor
        is multiplied by 0 (include/vqhip.h, vqhip_cvq_rows).  Same codebooks bit for bit on finite data; the column argmin
        runs over a fraction of the codebook and ONE all-reduce carries histogram, token count and [M, D] anchors instead
        of three collectives and a [K, D] tensor.  The list is built on the device from the synchronised probabilities, so
        every rank has the same one.  Eager steps size their exchange from a count that was copied to pinned memory at the
        end of the PREVIOUS step: the one host wait of a step is on the event of that copy, queued a whole step earlier
        (it caps the host's run-ahead at one step; a first step, or probabilities replaced from outside, count on the spot);
        under HIP-graph capture the launches are sized for K and the device-side count decides.  Automatic = NearestAnchor without sync, fused distance, D with a proposal image.
        False: the reference's dense data flow (a [K, D] anchor tensor, all-reduced on its own)."""
        super().__init__(*args, **kwargs)
        self._anchor = anchor
        self._eps = eps
        self._sparse_anchors = sparse_anchors
        self._listed = None               # (p tensor, its _version, rows, slot, count, pinned host count, copy event)
        self._pinned_count = None
        self._step_state = None           # train_step.CvqStepState of the one-call forward (lazily, on the codebook's device)
        # How a HIP-graph capture of the one-call forward sizes and chains itself (graphs.GraphedQuantizer sets it around its
        # captures; None = launches sized for K, the list rebuilt inside the graph): dict(cap=<capacity of the listed-code
        # launches>, chained=True: the graph trusts rows / slot / count to describe the probabilities it starts from — every
        # replay ends by writing the NEXT step's list there, and publishes its length to a pinned host word as soon as the step's
        # histogram is final — so the host can pick, per step, the smallest captured capacity that fits)
        self.capture_plan = None
        self.last_exchange_rows = None    # M of the last training step (diagnostics: bench.py, tests)

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
        self._listed = None               # a list prefetched for the old probabilities is void (`_prefetch_listed` re-arms it)
        if self._step_state is not None:
            self._step_state.invalidate()
        if self.quantizer.inplace_updates and '_probability' in self.quantizer._buffers \
                and self.quantizer._buffers['_probability'].shape == value.shape \
                and self.quantizer._buffers['_probability'].device == value.device:
            self.quantizer._buffers['_probability'].copy_(value)
        else:
            self.quantizer.register_buffer('_probability', value)

    # ---- anchors for the codes that can need one ------------------------------------------------------------------
    def _sparse_ok(self, d, hist32) -> bool:
        if self._sparse_anchors is False or type(self._anchor) is not NearestAnchor:
            return False
        if get_world_size() > exchange.MAX_WORLD:                # count pieces no longer exact in fp32: the dense flow below
            return False
        if self._sync_exchange() and d.shape[0] > ops.SYNC_MAX_ROWS:      # a key holds the row in 24 bits (include/vqhip.h)
            return False
        p = self.probability
        return (isinstance(d, LazyDistance) and hist32 is not None and p.is_cuda and p.dtype == torch.float32
                and ops.coarse_supported(self.quantizer.embedding_dim))

    def _sync_exchange(self) -> bool:
        """NearestAnchor(sync=True) over more than one rank: the GLOBAL nearest latent per code (anchors.py:50-57).  The reference
        all-gathers latents and the [N, K] matrix; here the ranks agree on the winner by a MIN all-reduce of 8-byte keys and the
        winner alone contributes its row to the packed SUM (include/vqhip.h: vqhip_cvq_col_keys)."""
        return bool(self._anchor._sync) and exchanging()

    def _listed_codes(self, p: torch.Tensor, K: int, capturing: bool):
        """(rows, slot, count, cap): the device-side list for the step that starts from ``p`` and a host-known bound on its
        length.  Eager: the list and its length were produced at the end of the previous step (`_prefetch_listed`); a
        first step, or a ``p`` changed from outside, computes them now (one synchronisation).  Capture: sized for K."""
        if capturing:
            rows, slot, count = ops.cvq_rows(p, K, self._ema.decay, self._eps)
            return rows, slot, count, K
        st = self._listed
        if st is not None and st[0] is self.probability and st[1] == st[0]._version and p.data_ptr() == st[0].data_ptr():
            st[6].synchronize()                                   # the copy was queued a whole step ago
            return st[2], st[3], st[4], int(st[5][0])
        rows, slot, count = ops.cvq_rows(p, K, self._ema.decay, self._eps)
        return rows, slot, count, int(count.item())

    def _prefetch_listed(self, p_new: torch.Tensor, K: int) -> None:
        rows, slot, count = ops.cvq_rows(p_new, K, self._ema.decay, self._eps)
        if self._pinned_count is None:                   # one pinned word for the life of the callback
            self._pinned_count = torch.empty(1, dtype=torch.int32).pin_memory()
        host = self._pinned_count
        host.copy_(count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._listed = (p_new, p_new._version, rows, slot, count, host, ev)

    def _sparse_step(self, x: torch.Tensor, quant: torch.Tensor, d: LazyDistance, hist32: torch.Tensor) -> None:
        K = self.quantizer.codebook_size
        world = get_world_size()
        weight = self.vector_quantizer.embedding.weight
        w_in, p_in = weight.detach().contiguous(), self.probability.contiguous()
        capturing = torch.cuda.is_current_stream_capturing()
        rows, slot, count, cap = self._listed_codes(p_in, K, capturing)
        self.last_exchange_rows = cap
        xr = x.detach()
        col = None
        if cap > 0:
            xq, eq = (d._xq, d._eq) if d._xq is not None and d._eq is not None else d._distance.exact_operands(d._x, d._e, d.metric)
            col = ops.col_argmin_rows(xq, eq, rows, count, cap, d.metric)
        inplace = self.quantizer.inplace_updates
        w_out = w_in if inplace else torch.empty_like(w_in)
        p_out = p_in if inplace else torch.empty_like(p_in)
        if not exchanging():
            ops.cvq_apply(w_in, w_out, p_in, p_out, slot, self._ema.decay, self._eps, hist32=hist32, numel=quant.numel(),
                          x=xr, col_idx=col, cap=cap)
        elif self._sync_exchange():                              # global winners: keys MIN-reduced, then the masked packed SUM
            packed = exchange.cvq_exchange_sync(hist32, quant.numel(), xr, xq if cap > 0 else None, eq if cap > 0 else None, rows,
                                                col, count, cap, K, d.metric)
            ops.cvq_apply(w_in, w_out, p_in, p_out, slot, self._ema.decay, self._eps, packed=packed, world=1, cap=cap)
        else:                                                    # histogram ‖ token count ‖ [cap, D] anchors: one all-reduce
            packed = exchange.cvq_exchange(hist32, quant.numel(), xr, col, count, cap, K)
            ops.cvq_apply(w_in, w_out, p_in, p_out, slot, self._ema.decay, self._eps, packed=packed, world=world, cap=cap)
        if inplace:
            if p_out is not self.probability:                    # (a non-contiguous buffer was copied above)
                self.probability.copy_(p_out)
            if w_out.data_ptr() != weight.data_ptr():
                weight.data.copy_(w_out)
            if Store.DRY_RUN:
                assert is_sync(w_out)
            self.vector_quantizer.invalidate_codebook()
        else:
            self._update_probability(p_out)
            self._update_embedding(w_out)
        if self._step_state is not None:                         # the one-call forward's list described the old probabilities
            self._step_state.invalidate()
        if not capturing:
            self._prefetch_listed(self.probability, K)

    # ---- the whole training forward as ONE library call (train_step.py, include/vqhip.h: vqhip_cvq_forward) --------------
    def fused_forward_ok(self, x: torch.Tensor) -> bool:
        """True when this step can be enqueued by one call: the conditions of the sparse-anchor flow (`_sparse_ok`) that can
        be known before the encode, a distance whose encode is the library's fused one, device latents."""
        from .distances import CosineDistance, L2Distance
        q = self.vector_quantizer
        if not (q.training and self.with_ema and x.dim() == 2 and x.is_cuda and x.shape[0] > 0 and x.shape[0] < (1 << 31)):
            return False
        if self._sparse_anchors is False or type(self._anchor) is not NearestAnchor:
            return False
        if get_world_size() > exchange.MAX_WORLD or (self._sync_exchange() and x.shape[0] > ops.SYNC_MAX_ROWS):
            return False
        if type(q.distance) not in (L2Distance, CosineDistance) or not ops.coarse_supported(q.embedding_dim) or q._cache_codebook:
            return False
        if '_probability' not in q._buffers:
            return False
        p, w = self.probability, q.embedding.weight
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == x.device
                and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous())

    def refresh_list(self) -> None:
        """(Re)build the device-side list of codes for the CURRENT probabilities and note its length on the host — what a
        first step, a loaded checkpoint or a probability buffer replaced from outside needs (one synchronisation)."""
        from .. import train_step
        p = self.probability
        K = self.quantizer.codebook_size
        if self._step_state is None or self._step_state.device != p.device or self._step_state.K != K:
            self._step_state = train_step.CvqStepState(K, p.device)
        st = self._step_state
        ops.cvq_rows(p, K, self._ema.decay, self._eps, out=(st.rows, st.slot, st.count))
        st.count_host[0] = int(st.count.item())
        st.mark_list(p)

    def fused_forward(self, x: torch.Tensor, memo: Memo, beta: float):
        """_encode + after_encode (sparse-anchor flow of `_sparse_step`) + decode + MSE losses + STE of one training step from
        one host call.  Returns (quant, z_ste, m_cb, m_cm, m_vqgan) with the reference's memo side effects."""
        from .. import functional as VF, train_step
        from ..utils import all_reduce_min, all_reduce_sum, get_rank
        q = self.vector_quantizer
        weight = q.embedding.weight
        K, D = weight.shape
        capturing = torch.cuda.is_current_stream_capturing()
        p_in = self.probability
        if self._step_state is None or self._step_state.device != p_in.device or self._step_state.K != K:
            self._step_state = train_step.CvqStepState(K, p_in.device)
        st = self._step_state
        if capturing:                     # the launches are sized for a fixed capacity, the device-side count decides
            plan = self.capture_plan or {}
            list_ready = prefetch = early = bool(plan.get('chained', False))
            cap = int(plan.get('cap', K))
        else:
            if not st.list_valid_for(p_in):
                self.refresh_list()       # first step / probabilities replaced from outside: counted on the spot
                st = self._step_state
                list_ready, cap = True, int(st.count_host[0])
            else:
                list_ready, cap = True, -1          # the library reads the count the previous step's prefetch copied out
            prefetch, early = True, False
        inplace = q.inplace_updates
        w_inplace = self._writes_in_place(weight)
        w_in = weight.detach()
        w_out = w_in if w_inplace else torch.empty_like(w_in)
        p_out = p_in if inplace else torch.empty_like(p_in)
        exch = exchanging()
        sync = self._sync_exchange()
        comm = None
        if exch and not exchange_log.enabled:
            from .. import rccl
            comm = rccl.communicator(w_in)
        metric = q.distance.metric_for(D)
        xd = x.detach()
        # the codebook operand of memo['distance'] aliases the storage the encode ran against (quantizers.py:97 clones it)
        e_alias = weight.view_as(weight) if (torch.is_grad_enabled() and weight.requires_grad) else w_in
        out = train_step.cvq_forward(xd, w_in, p_in, w_out, p_out, metric, self._ema.decay, self._eps, beta, st, cap=cap,
                                     list_ready=list_ready, prefetch=prefetch, exchange=exch, world=get_world_size(), comm=comm,
                                     all_reduce=all_reduce_sum if exch else None, tail=True, early_count=early,
                                     anchor_sync=sync, rank=get_rank() if sync else 0, all_reduce_min=all_reduce_min if sync else None)
        self.last_exchange_rows = out['cap_used']
        self._listed = None                                      # (the hook-by-hook flow's prefetched list is void now)
        if Store.DRY_RUN:
            assert is_sync(w_out)
        p_new = p_in
        if not inplace:
            self._update_probability(p_out)
            p_new = p_out
        if not w_inplace:
            weight.data = w_out                                  # callbacks/update.py:56
        q.invalidate_codebook()
        if prefetch:
            st.mark_list(p_new)                                  # rows / slot / count now describe p_new (the call's last but one launch)
        enc = get_memo(memo, 'encode')
        prepared = out['prepared']
        cos = out['xq'] is not None
        x_op = x if (torch.is_grad_enabled() and x.requires_grad) else out['x']
        enc['distance'] = LazyDistance(q.distance, x_op, e_alias, xq=out['xq'] if cos else out['x'],
                                       eq=prepared.exact_rows() if cos else w_in, metric=ops.metric_name(prepared.metric))
        enc['hist'] = out['hist']
        memo['encode'] = enc
        done = VF._Computed(z_ste=out['z_ste'], mse=out['mse'], idx=out['idx'])
        z_ste, m_cb, m_cm, m_vqgan = VF.precomputed_decode_loss(x, weight, done, beta)
        return out['idx'], z_ste, m_cb, m_cm, m_vqgan

    def after_encode(self, x: torch.Tensor, quant: torch.Tensor, memo: Memo) -> torch.Tensor:
        quant = super().after_encode(x, quant, memo)
        if not self.quantizer.training:
            return quant
        K = self.quantizer.codebook_size
        d = memo['encode']['distance']
        hist32 = memo['encode'].get('hist')
        if self._sparse_ok(d, hist32):
            self._sparse_step(x, quant, d, hist32)
            return quant
        if (not exchanging() and type(self._anchor) is NearestAnchor and not self._anchor._sync
                and isinstance(d, LazyDistance) and hist32 is not None
                and self.probability.is_cuda and self.probability.dtype == torch.float32):
            # one rank, dense form: the whole update in one launch on the epilogue histogram, the column argmin and the
            # latents (vqhip_cvq_step; bit-identical to the staged form below)
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
        # the reference's data flow: statistics exchange, probability update, anchor sampler with its own exchange, blend
        e = self.quantizer.embeddings
        stats = QuantStatistics(quant=quant, codebook_size=K, sync=True, hist=memo['encode'].get('hist'))
        hist, numel = stats.bin_count(), stats._statistics()[1]
        p = self.probability.to(device=e.device, dtype=torch.float32).clone()
        ops.cvq_update_(e, p, hist, numel, None, self._ema.decay, self._eps, stage=1)      # p = ema(p, hist/numel)
        self._update_probability(p)
        anchors, memo = self._anchor(x.detach(), e, d, quant, p, memo=memo)
        # decay = 1 - exp(-p*K*10/(1-ema.decay) - eps); e = e*decay + anchors*(1-decay)
        ops.cvq_update_(e, p, None, None, anchors, self._ema.decay, self._eps, stage=2)
        self._update_embedding(e)
        return quant
