"""Small host-side helpers standing in for the todd utilities the path uses:
EMA / ema, PriorityQueue, Store flags, rank helpers, is_sync, and the packed statistics all-reduce."""
from __future__ import annotations

import os
from typing import Generic, Iterable, Mapping, Optional, TypeVar

import torch
import torch.distributed as dist

T = TypeVar('T')


# ---- todd.utils.ema / EMA (definition fixed in SURVEY.md §8c: a*decay + b*(1-decay), default decay 0.99) ----

def ema(a: torch.Tensor, b: torch.Tensor, decay) -> torch.Tensor:
    return a * decay + b * (1 - decay)


class EMA:
    def __init__(self, *args, decay: float = 0.99, **kwargs) -> None:
        self._decay = float(decay)

    @property
    def decay(self) -> float:
        return self._decay

    def __call__(self, a: Optional[torch.Tensor], b: torch.Tensor) -> torch.Tensor:
        if a is None:
            return b
        return ema(a, b, self._decay)


# ---- todd.runners.utils.PriorityQueue: per-hook ordering of callbacks (ascending priority, stable) ----

class PriorityQueue(Generic[T]):
    def __init__(self, priorities: Iterable[Mapping[str, int]], items: Iterable[T]) -> None:
        self._priorities = [dict(p) for p in priorities]
        self._items = list(items)
        assert len(self._priorities) == len(self._items)

    def __call__(self, key: str) -> list:
        order = sorted(range(len(self._items)), key=lambda i: (self._priorities[i].get(key, 0), i))
        return [self._items[i] for i in order]

    def __len__(self) -> int:
        return len(self._items)


# ---- todd.Store flags (vq/utils/stores.py:8-10): environment-backed booleans ----

class _Store:
    @property
    def DRY_RUN(self) -> bool:  # noqa: N802
        return bool(os.environ.get('DRY_RUN'))


Store = _Store()


# ---- rank helpers (todd.patches.torch.get_rank / get_world_size) ----

def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def exchanging() -> bool:
    """True when the codebook update has an exchange step to run: a process group of more than one rank — or, with
    ``VQ_FORCE_EXCHANGE=1``, a process group of ANY size.  The forced world-size-1 group sends the update through the whole
    multi-rank flow (pack, the RCCL collective, apply) on the single GPU a test box has; the results are those of the
    one-rank flow bit for bit (tests/test_gpu_rccl.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('VQ_FORCE_EXCHANGE') == '1'


# ---- collectives that also work on a backend without device-tensor support for the op (gloo: only broadcast and
#      all_reduce take device tensors) — staged through the host there; RCCL ("nccl") takes the device tensors directly ----

def _stage(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() == 'gloo'


def all_gather(t: torch.Tensor) -> list:
    """[t_rank0, t_rank1, ...] (todd.patches.torch.all_gather; same shape on every rank)."""
    src = t.contiguous().cpu() if _stage(t) else t.contiguous()
    out = [torch.empty_like(src) for _ in range(get_world_size())]
    dist.all_gather(out, src)
    return [o.to(t.device) for o in out] if _stage(t) else out


def gather_to_rank0(t: torch.Tensor):
    """torch.distributed.gather to rank 0: the list of every rank's tensor on rank 0, None elsewhere."""
    src = t.contiguous().cpu() if _stage(t) else t.contiguous()
    if get_rank() > 0:
        dist.gather(src)
        return None
    out = [torch.zeros_like(src) for _ in range(get_world_size())]
    dist.gather(src, out)
    return [o.to(t.device) for o in out] if _stage(t) else out


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    dist.broadcast(t, src)          # device tensors are fine on gloo and RCCL alike
    return t


def is_sync(t: torch.Tensor) -> bool:
    """True when ``t`` is bit-identical on every rank (todd.utils.is_sync; the reference's only
    distributed-correctness assert: callbacks/update.py:54-55, cvqvae/anchors.py:52-53,62-63)."""
    if get_world_size() <= 1:
        return True
    flat = t.detach().contiguous().view(torch.uint8).reshape(-1) if t.dtype != torch.bool else t.reshape(-1)
    lo, hi = flat.clone().to(torch.int32), flat.clone().to(torch.int32)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


# ---- accounting of the data-path collectives (bench.py: exchange_bytes_per_step, collective_ms) ----

class _ExchangeLog:
    """Bytes, call count and (optionally) device time of the codebook-update collectives.  Off by default: one attribute
    test per collective.  ``timing=True`` brackets every collective with events on the current stream (the collective
    itself runs on the backend's stream; the second event waits for it)."""

    def __init__(self) -> None:
        self.enabled = False
        self.timing = False
        self.reset()

    def reset(self) -> None:
        self.calls = 0
        self.bytes = 0
        self._events = []

    def start(self, timing: bool = False) -> None:
        self.reset()
        self.enabled, self.timing = True, timing

    def stop(self) -> dict:
        ms = None
        if self.timing and self._events:
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in self._events)
        out = dict(calls=self.calls, bytes=self.bytes, ms=ms)
        self.enabled = self.timing = False
        self._events = []
        return out

    def collective(self, fn, tensor: torch.Tensor, *args, **kwargs):
        if not self.enabled:
            return fn(tensor, *args, **kwargs)
        self.calls += 1
        self.bytes += tensor.numel() * tensor.element_size()
        if self.timing and tensor.is_cuda and not torch.cuda.is_current_stream_capturing():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(tensor, *args, **kwargs)
            b.record()
            self._events.append((a, b))
            return out
        return fn(tensor, *args, **kwargs)


exchange_log = _ExchangeLog()


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of one tensor, in place, counted by ``exchange_log``.  An fp32 device tensor on an RCCL process group
    goes through ``vqhip_allreduce_packed`` — the collective enqueued on the current (compute) stream — anything else, and
    every gloo group, through ``dist.all_reduce`` (rccl.py)."""
    from . import rccl
    if os.environ.get('VQ_DEBUG_SKIP_ALLREDUCE') == '1' and get_world_size() == 1:      # measurement aid: a one-rank SUM is the identity
        return t
    comm = rccl.communicator(t)
    if comm is not None:
        exchange_log.collective(rccl.all_reduce, t, comm)
    else:
        exchange_log.collective(dist.all_reduce, t)
    return t


def all_reduce_min(t: torch.Tensor) -> torch.Tensor:
    """MIN all-reduce of an int64 tensor, in place, counted by ``exchange_log`` — the key exchange of NearestAnchor(sync=True)
    (include/vqhip.h: vqhip_cvq_col_keys).  Routes as ``all_reduce_sum``."""
    from . import rccl
    if os.environ.get('VQ_DEBUG_SKIP_ALLREDUCE') == '1' and get_world_size() == 1:
        return t
    comm = rccl.communicator(t, dtype=torch.int64)
    if comm is not None:
        exchange_log.collective(rccl.all_reduce_min, t, comm)
    else:
        exchange_log.collective(dist.all_reduce, t, op=dist.ReduceOp.MIN)
    return t


# ---- one exchange step of the codebook update ----

def all_reduce_statistics(hist: torch.Tensor, numel: Optional[int] = None, sums: Optional[torch.Tensor] = None):
    """SUM-all-reduce the per-step codebook statistics with at most two collectives:
    int64 [K+1] = histogram ‖ token count (exact), and fp32 [K*D] = per-code sums (centroids / anchors).
    Replaces the reference's separate all_reduce calls (vq/algorithms/vq/utils.py:35 twice,
    vqkd/quantizers/callbacks.py:63-64, cvqvae/anchors.py:65-67).  No-op for a single rank.
    Returns (hist int64[K], numel, sums); numel is a python int for one rank and a device int64 scalar tensor after
    an all-reduce (no host synchronisation)."""
    K = hist.numel()
    if numel is None:
        numel = int(hist.sum().item()) if get_world_size() <= 1 else None
    if get_world_size() <= 1:
        return hist.to(torch.int64), numel, sums
    packed = torch.empty(K + 1, dtype=torch.int64, device=hist.device)
    packed[:K] = hist
    packed[K] = numel if numel is not None else hist.sum()
    all_reduce_sum(packed)
    if sums is not None:
        if not sums.is_contiguous():
            sums = sums.contiguous()
        all_reduce_sum(sums)
    return packed[:K], packed[K], sums
