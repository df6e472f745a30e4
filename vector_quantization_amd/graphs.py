"""HIP-graph replay of a quantizer step for launch-bound shapes (training batches of a few thousand tokens).

At the reference's per-rank training batch (12 images = 3072 tokens, K = 16384, D = 256) the step is about forty short
kernels; issued eagerly from Python the GPU idles between them (measured: 0.18-0.35 ms of kernels inside a 0.7-0.9 ms
step).  ``GraphedQuantizer`` captures the whole step — callbacks with their codebook update, decode, losses and, in train
mode, the backward — once for a fixed input shape and replays it with one launch per step.

What capture needs from the step, and how the quantizer provides it:
  * no host synchronisation and no allocation outside torch's graph pool — true of every libvqhip entry point
    (include/vqhip.h) and of the module code on the default paths;
  * stable addresses: the reference's callbacks REBIND ``weight.data`` / re-register ``_probability`` with a fresh tensor
    every step (callbacks/update.py:56, quantizer_callback.py:72-73); a replayed graph would keep reading the old
    storage.  ``quantizer.inplace_updates = True`` makes the same callbacks write the same values INTO the existing
    storage instead (value-identical; optimizer state keyed on the Parameter is unaffected either way).
  * a codebook image that follows the weight: ``cache_codebook`` is switched off (its Python-side key would be evaluated
    once, at capture), so the image kernels are part of the graph;
  * ``memo['encode']['distance']`` of a replayed step is not available (the step returns tensors only); with
    ``inplace_updates`` its codebook operand is the live storage, i.e. a consumer that materialises the matrix AFTER the
    update callback sees the updated codebook (the reference clones: quantizers.py:97) — no shipped config does.
Collectives: captured as issued; RCCL ("nccl") supports capture, gloo does not — use the graph at world size 1 or on RCCL.

CVQ-VAE (the sparse-anchor flow): a captured step cannot size its listed-code launches from a count the host reads, so the
train-mode step is captured at SEVERAL capacities (128, 1024, 4096, 8192 and K listed codes) and chained: every replay ends by writing the
next step's list, and publishes that list's LENGTH — with a sequence number, to a pinned host word — as soon as the step's
histogram is final (one small launch behind the encode / the exchange: the probabilities' update needs nothing else); before the
next replay the host polls the word for the number it expects and picks the smallest captured capacity that fits, while the
GPU is still busy with the rest of the previous replay (round 4 captured K only: at
65 536 tokens per rank the replayed step was slower than the eager one, 1.08 against 0.79 ms, profiles/r04_cvq256.json).
"""
from __future__ import annotations

import gc
import time
from typing import Optional

import torch
from torch import nn


def _nccl_group() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'


def _quiesce_process_group() -> None:
    """ProcessGroupNCCL's watchdog thread polls the events of the collectives it has been handed; while a stream captures in
    the default (global) capture mode an event query from ANOTHER thread is an error, and the watchdog answers it by aborting
    the process (observed as a sporadic SIGABRT of the rank: rounds 4-6, `Watchdog::run()` in the backtrace).  Eager collectives
    issued shortly before a capture are still on its list although they have completed.  Before a capture: everything
    synchronised, then a few of the watchdog's 100-ms sweeps to let it retire its list."""
    if _nccl_group():
        torch.cuda.synchronize()
        time.sleep(0.35)


def _graphed_train_callable(st: nn.Module, sample: torch.Tensor, warmup: int):
    """torch.cuda.make_graphed_callables(st) — with the warm-up iterations run HERE on an nccl process group: their (eager)
    collectives must have left ProcessGroupNCCL's watchdog before the capture begins (`_quiesce_process_group`); the helper's own
    warm-up runs them right in front of it."""
    if not _nccl_group():
        return torch.cuda.make_graphed_callables(st, (sample,), num_warmup_iters=warmup, allow_unused_input=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    params = tuple(p for p in st.parameters() if p.requires_grad)
    with torch.cuda.stream(side):
        for _ in range(warmup):
            outs = tuple(o for o in st(sample) if isinstance(o, torch.Tensor) and o.requires_grad)
            if outs:
                torch.autograd.grad(outs, (sample,) + params, grad_outputs=tuple(torch.empty_like(o) for o in outs), allow_unused=True)
    torch.cuda.current_stream().wait_stream(side)
    _quiesce_process_group()
    return torch.cuda.make_graphed_callables(st, (sample,), num_warmup_iters=0, allow_unused_input=True)


class _Step(nn.Module):
    """forward(x) -> (z, loss): the quantizer call with a fresh memo, tensors only (what graph capture wants).  The tokens
    are not a differentiable output: handed back through ``make_graphed_callables`` they would make autograd materialise a
    zero "gradient" for them at every backward (one more fill kernel per step); they are kept as ``last_quant`` instead — the
    captured step's own output tensor (graph pool memory, kept alive by this reference), rewritten by every replay."""

    def __init__(self, quantizer: nn.Module) -> None:
        super().__init__()
        self.quantizer = quantizer
        self.last_quant: Optional[torch.Tensor] = None

    def forward(self, x: torch.Tensor):
        z, loss, memo = self.quantizer(x, {})
        self.last_quant = memo['quant']
        return z, loss


class GraphedQuantizer(nn.Module):
    """``z, loss, quant = GraphedQuantizer(q, sample_x)(x)`` — same values as ``q(x, {})`` step after step, replayed from
    HIP graphs.  ``x`` must have the sample's shape and dtype.  Train mode (``q.training``) captures forward + backward
    with ``torch.cuda.make_graphed_callables``; eval mode captures the forward under ``no_grad``.

    Warm-up and capture run the step a few times on ``sample_x``; parameters and buffers are restored afterwards, so the
    codebook is exactly what it was before the constructor ran.  (A VQ-KD quantizer's lazy k-means init, which needs host
    logic, must already have happened: call the quantizer once eagerly first.)"""

    def __init__(self, quantizer: nn.Module, sample_x: torch.Tensor, warmup: int = 3, bucket_caps=(128, 1024, 4096, 8192)) -> None:
        super().__init__()
        if not sample_x.is_cuda:
            raise ValueError('GraphedQuantizer needs a device tensor (no CPU path)')
        if len(getattr(quantizer, '_forward_pre_hooks', {})) > 0:
            raise RuntimeError('the quantizer still has a pending forward pre-hook (lazy init): run one eager step first')
        self.quantizer = quantizer
        quantizer.inplace_updates = True
        # A cached codebook image is chosen by a Python-side key at CAPTURE time: the prepare launch would not be recorded
        # and every replay would propose candidates from the image frozen then, whatever the weight has become.  The graph
        # therefore always contains the image kernels (17 us at K = 16 384, D = 256) and follows the live weight.
        if getattr(quantizer, '_cache_codebook', False):
            quantizer._cache_codebook = False
        if hasattr(quantizer, 'invalidate_codebook'):
            quantizer.invalidate_codebook()
        # scratch of the fused loss mean, owned by this graph (ops.owned_mse_scratch)
        self._mse_scratch = torch.zeros(16, dtype=torch.uint8, device=sample_x.device)
        self._train = quantizer.training
        self._shape, self._dtype = tuple(sample_x.shape), sample_x.dtype
        saved = {k: v.detach().clone() for k, v in quantizer.state_dict().items()}
        step = self._step = _Step(quantizer)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._caps, self._calls, self._steps = [], [], []
        self.poll_timeout_s = 2.0                # how long a replay waits for the previous one's list length before it counts itself
        self._cvq = self._chained_cvq_callback(quantizer, sample_x) if self._train else None
        from . import ops
        # No cyclic garbage collection while anything captures: an older graph that has become garbage (a GraphedQuantizer the caller
        # dropped) would be destroyed wherever the collector happens to run — hipGraphExecDestroy during another stream's capture is
        # "operation not permitted when stream is capturing", raised inside a destructor: the process is terminated (seen as a
        # sporadic SIGABRT of tests that build several graphed quantizers in a row, round 6).  Collect now, then hold the collector.
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_all(quantizer, sample_x, warmup, bucket_caps, step, ops)
        finally:
            if gc_was_enabled:
                gc.enable()
        with torch.no_grad():                                    # undo the side effects of warm-up and capture
            for k, v in quantizer.state_dict().items():
                v.copy_(saved[k])

    def _capture_all(self, quantizer, sample_x, warmup, bucket_caps, step, ops) -> None:
        with ops.owned_mse_scratch(self._mse_scratch):
            if self._cvq is not None:
                K = quantizer.codebook_size
                self._caps = sorted({int(c) for c in bucket_caps if 0 < int(c) < K}) + [K]        # ascending, no duplicates
                try:
                    for cap in self._caps:
                        self._cvq.capture_plan = dict(cap=cap, chained=True)
                        st = _Step(quantizer)
                        sample = sample_x.detach().clone().requires_grad_(True)
                        self._calls.append(_graphed_train_callable(st, sample, warmup))
                        self._steps.append(st)
                finally:
                    self._cvq.capture_plan = None
                self._list_version = None            # the chained list is (re)built before the first replay
                self._seq = None                     # sequence number the last launched replay will publish (set at the first one)
            elif self._train:
                sample = sample_x.detach().clone().requires_grad_(True)
                # (allow_unused_input: a VQ-KD step gives the codebook no gradient — the commitment term and the straight-through
                #  output both detach z, configs/vqkd/model.py:76-82 freezes the quantizer anyway — and the one-call forward's
                #  autograd node says so with None instead of a zero-filled [K, D] tensor)
                self._call = _graphed_train_callable(step, sample, warmup)
            else:
                self._x = sample_x.detach().clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(warmup):
                        step(self._x)
                torch.cuda.current_stream().wait_stream(side)
                _quiesce_process_group()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph), torch.no_grad():
                    z, loss = step(self._x)
                    self._out = (z, loss, step.last_quant)

    @staticmethod
    def _chained_cvq_callback(quantizer: nn.Module, sample_x: torch.Tensor):
        """The CVQVAECallback of a quantizer whose train step is the one-call CVQ-VAE forward (capacity buckets), else None."""
        step = getattr(quantizer, '_one_call_step', None)
        if step is None or sample_x.dim() != 2:
            return None
        bound = step(sample_x)
        if bound is None or getattr(bound, '__name__', '') != '_forward_cvq':
            return None
        return quantizer._callbacks.callbacks[0]

    def _replay_chained(self, x: torch.Tensor):
        cb = self._cvq
        p = cb.probability
        st = cb._step_state
        me = ('replay', id(self))
        # The chain trusts rows / slot / count / the early word to describe the probabilities the replay starts from.  That holds
        # only while the LAST writer of that state was a replay of this object: an eager train step in between (a ragged last
        # batch, say) rewrites the list in place without touching the early word or `p._version`, and so does another
        # GraphedQuantizer on the same module — `writer` (train_step.CvqStepState) names the last one
        if st is None or self._seq is None or st.writer != me or self._list_version != (p._version, p.data_ptr()):
            count = self._resync()
            st = cb._step_state
        else:
            # the previous replay publishes {its sequence number, the length of THIS step's list} as soon as its histogram is final:
            # usually long since there — the rest of that replay (column pass, update, decode, backward) is what the GPU is running now
            word = int(st.early_host[0])
            if (word >> 32) != self._seq:
                deadline = time.monotonic() + self.poll_timeout_s
                while (word >> 32) != self._seq and time.monotonic() < deadline:
                    word = int(st.early_host[0])
            if (word >> 32) == self._seq:
                count = word & 0xFFFFFFFF
            else:                                                 # never published (a failed replay, a non-coherent pinned pool): count on the spot
                count = self._resync()
                st = cb._step_state
        which = next(i for i, cap in enumerate(self._caps) if cap >= count)
        self.last_capacity = self._caps[which]
        z, loss = self._calls[which](x)
        self._seq += 1                                            # every replay advances the device counter by one
        st.writer = me
        st.list_of = None                                         # (an eager step after this one rebuilds the list: the host knows no count)
        return z, loss, self._steps[which].last_quant

    def _resync(self) -> int:
        """Rebuild the list for the current probabilities and re-read the device's sequence number (one synchronisation): the first
        replay, probabilities changed from outside, or another writer since the last replay."""
        cb = self._cvq
        cb.refresh_list()
        st = cb._step_state
        p = cb.probability
        self._list_version = (p._version, p.data_ptr())
        self._seq = int(st.seq_dev.item())
        return int(st.count_host[0])

    def forward(self, x: torch.Tensor):
        if tuple(x.shape) != self._shape or x.dtype != self._dtype:
            raise ValueError(f'graph captured for {self._shape} {self._dtype}, got {tuple(x.shape)} {x.dtype}')
        if self._cvq is not None:
            return self._replay_chained(x)
        if self._train:
            z, loss = self._call(x)
            return z, loss, self._step.last_quant
        self._x.copy_(x)
        self._graph.replay()
        return self._out
