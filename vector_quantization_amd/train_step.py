"""One library call per training forward (include/vqhip.h: vqhip_cvq_forward, vqhip_vqkd_forward).

The reference's training step runs the quantizer forward as ~20 ATen calls; re-hosted call for call on libvqhip an eager
nn.Module step still makes ~10 host calls, and at the reference's per-rank batches (3 072 - 12 544 tokens) the HOST is the
bound (profiles/r05_train_shapes_before.txt: the host has issued a step exactly when the GPU has finished it).  Here the
whole forward of the two callback-driven training configs is enqueued by ONE call:

  CVQ-VAE (vq/algorithms/cvqvae/quantizer_callback.py:75-105): encode -> listed codes -> column argmin -> [pack ->
      all-reduce] -> apply -> prefetch of the next step's list -> decode / straight-through / loss
  VQ-KD   (vq/algorithms/vqkd/quantizers/callbacks.py:114-129): normalise codebook (twice) and latents -> encode ->
      histogram + centroid sums into the packed buffer -> [all-reduce] -> EMA update -> decode / STE / normalised MSE

Values are those of the chain of separate calls (the library makes the same launches in the same order).  This module owns
the persistent device buffers such a call needs (the list of codes, the pinned count word and its event, the arena) and the
argument blocks; the callbacks decide WHEN the fused form applies and keep the reference's memo side effects."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib, ops
from ._lib import STEP_AFTER_EXCHANGE, STEP_ALL, STEP_BEFORE_EXCHANGE, STEP_PACK_SYNC, check
from .ops import METRICS, _bytes, _codebook, _latents, _mse_scratch, _on_tensor_device, _stream


def _p(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


class _Arena:
    """Workspace + exchange buffer of one (shape, stream): allocated once, reused by every step on that stream (a fresh
    torch allocation per step costs host time, and the sizes never change in a training run).  Under HIP-graph capture
    nothing persistent is created: the buffers of a captured call come from the graph's pool."""

    def __init__(self) -> None:
        self._bufs: dict = {}

    def get(self, key, ws_bytes: int, packed_floats: int, device):
        if torch.cuda.is_current_stream_capturing():
            return _bytes(ws_bytes, device), (torch.empty(packed_floats, dtype=torch.float32, device=device) if packed_floats else None)
        skey = (key, ops._raw_stream(device.index) if ops._raw_stream is not None else 0)
        hit = self._bufs.get(skey)
        if hit is None or hit[0].numel() < ws_bytes or (packed_floats and (hit[1] is None or hit[1].numel() < packed_floats)):
            if len(self._bufs) > 8:            # shapes come and go (variable last batches): keep the set small
                self._bufs.clear()
            hit = (_bytes(ws_bytes, device), torch.empty(packed_floats, dtype=torch.float32, device=device) if packed_floats else None)
            self._bufs[skey] = hit
        return hit


class CvqStepState:
    """What a CVQVAECallback keeps between fused steps: the device-side list of the codes that can need an anchor
    (vqhip_cvq_rows) for the CURRENT probabilities, its length in a pinned host word with the event of the copy, and the
    arena.  ``list_of`` names the probability tensor (object, version, storage) the list was made from."""

    def __init__(self, K: int, device: torch.device) -> None:
        self.K, self.device = K, device
        self.rows = torch.empty(K, dtype=torch.int32, device=device)
        self.slot = torch.empty(K, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        with torch.cuda.device(device):
            self.event.record()                                   # created now: the library records / waits on its handle
        self.event_handle = int(getattr(self.event, 'cuda_event', 0) or 0)
        # early count (include/vqhip.h, vqhip_cvq_forward_t.early_word_host): {sequence number << 32 | count}, and the device counter
        self.early_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.seq_dev = torch.zeros(1, dtype=torch.int32, device=device)
        # the prefetched count reaches the host through a device store into pinned memory read behind an event: only where pinned
        # host memory is coherent (hipHostMalloc's default).  HIP_HOST_COHERENT=0 switches that off process-wide: the eager step then
        # counts on the spot (`CVQVAECallback.refresh_list`: one synchronisation per step) instead of trusting the word
        self.host_word_coherent = os.environ.get('HIP_HOST_COHERENT', '1') != '0'
        self.list_of = None
        self.writer = None        # who wrote rows / slot / count last: 'eager' (a host-known count behind `event`) or a GraphedQuantizer's token
        self.keys = None                                          # int64 [K]: NearestAnchor(sync=True)'s key exchange (lazily)
        self.arena = _Arena()

    def list_valid_for(self, p: torch.Tensor) -> bool:
        """True when rows / slot / count AND the pinned count word behind `event` describe ``p``: the last writer was an eager
        step (or `refresh_list`) for exactly this tensor.  A graph replay writes the list in place without a host-side record —
        `writer` then names the replaying object and the next eager step recounts."""
        lo = self.list_of
        return (self.host_word_coherent and self.writer == 'eager' and lo is not None and lo[0] is p and lo[1] == p._version
                and lo[2] == p.data_ptr())

    def mark_list(self, p: torch.Tensor) -> None:
        self.list_of = (p, p._version, p.data_ptr())
        self.writer = 'eager'

    def invalidate(self) -> None:
        self.list_of = None
        self.writer = None


@_on_tensor_device
def cvq_forward(x: torch.Tensor, w_in: torch.Tensor, p_in: torch.Tensor, w_out: torch.Tensor, p_out: torch.Tensor, metric,
                ema_decay: float, eps: float, beta: float, state: CvqStepState, *, cap: int, list_ready: bool, prefetch: bool,
                exchange: bool, world: int, comm: Optional[int], all_reduce=None, tail: bool = True, early_count: bool = False,
                anchor_sync: bool = False, rank: int = 0, all_reduce_min=None):
    """The CVQ-VAE training forward as one library call (two around a caller-issued collective when ``comm`` is None and the
    exchange has more than one rank: ``all_reduce(packed_view)`` is then called between the halves).

    cap >= 0: capacity of the listed-code launches; cap < 0: the library reads the prefetched count itself (the pinned word
    of ``state``, behind the event of its copy).  ``anchor_sync`` (with ``exchange``): NearestAnchor(sync=True) — the ranks agree
    on the global nearest latent per listed code through ``all_reduce_min(keys)`` before the packed SUM (three library calls
    around the two collectives, one with a communicator).  Returns a dict: idx, hist, xq (cosine), prepared (the codebook image),
    z_ste, mse (fp32[4]), cap_used, exchange_floats."""
    ops._require_cuda(x, w_in, p_in, w_out, p_out)
    x, dt = _latents(x)
    N, D = x.shape
    K = w_in.shape[0]
    m = METRICS[metric]
    L = _lib.lib()
    dev = x.device
    cos = m in (_lib.METRIC_COS, _lib.METRIC_COS_BF16)
    capturing = torch.cuda.is_current_stream_capturing()
    cap_max = cap if cap >= 0 else K
    ws_bytes = L.vqhip_cvq_forward_ws_bytes(N, K, D, cap_max)
    ws, packed = state.arena.get((N, K, D, dt, cap_max), ws_bytes, ops.pack_floats(K, cap_max, D) if exchange else 0, dev)
    image = _bytes(L.vqhip_codebook_bytes(K, D), dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    hist = torch.empty(K, dtype=torch.int32, device=dev)
    xq = torch.empty(N, D, dtype=torch.float32, device=dev) if cos else None
    z_ste = torch.empty(N, D, dtype=torch.float32, device=dev) if tail else None
    mse = torch.empty(4, dtype=torch.float32, device=dev) if tail else None
    a = _lib.CvqForwardArgs()
    a.struct_bytes = ctypes.sizeof(_lib.CvqForwardArgs)
    a.N, a.K, a.D, a.x_dtype, a.metric, a.world = N, K, D, dt, m, int(world)
    a.ema_decay, a.eps, a.beta = float(ema_decay), float(eps), float(beta)
    a.exchange, a.list_ready, a.prefetch = int(bool(exchange)), int(bool(list_ready)), int(bool(prefetch))
    a.cap = int(cap)
    a.x, a.w_in, a.p_in, a.w_out, a.p_out = x.data_ptr(), w_in.data_ptr(), p_in.data_ptr(), w_out.data_ptr(), p_out.data_ptr()
    a.rows, a.slot, a.count = state.rows.data_ptr(), state.slot.data_ptr(), state.count.data_ptr()
    # the pinned count word: written by the prefetch (a store of the list kernel itself, capturable), read by the call when cap < 0;
    # the event around it belongs to eager steps only (a captured step's replay is followed by the caller's own event)
    use_host_word = prefetch or cap < 0
    a.count_host = state.count_host.data_ptr() if use_host_word else None
    a.count_event = (state.event_handle or None) if (use_host_word and not capturing) else None
    a.comm = comm
    a.cb, a.cb_bytes = image.data_ptr(), image.numel()
    a.idx, a.hist, a.xq = idx.data_ptr(), hist.data_ptr(), _p(xq)
    a.packed, a.packed_floats = _p(packed), (packed.numel() if packed is not None else 0)
    a.z_ste, a.mse = _p(z_ste), _p(mse)
    a.scratch16 = _mse_scratch(dev).data_ptr() if tail else None
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    a.cap_used, a.exchange_floats = -1, 0
    a.early_word_host = state.early_host.data_ptr() if early_count else None
    a.early_seq_dev = state.seq_dev.data_ptr() if early_count else None
    sync = bool(anchor_sync and exchange)
    a.anchor_sync, a.rank = int(sync), int(rank)
    keys = None
    if sync:
        if capturing:
            keys = torch.empty(K, dtype=torch.int64, device=dev)
        else:
            if state.keys is None:
                state.keys = torch.empty(K, dtype=torch.int64, device=dev)
            keys = state.keys
    a.keys = _p(keys)
    stream = _stream()
    if cap < 0 and not state.event_handle:           # no raw handle on this torch build: the wait happens here instead
        state.event.synchronize()
        a.cap = int(state.count_host[0])
    one_call = not exchange or comm is not None or (world == 1 and all_reduce is None)
    if one_call:
        a.phases = STEP_ALL
        check(L.vqhip_cvq_forward(ctypes.byref(a), stream), 'vqhip_cvq_forward')
    else:
        a.phases = STEP_BEFORE_EXCHANGE
        check(L.vqhip_cvq_forward(ctypes.byref(a), stream), 'vqhip_cvq_forward')
        if sync:
            if a.cap_used > 0:
                all_reduce_min(keys[:a.cap_used])
            a.phases = STEP_PACK_SYNC
            check(L.vqhip_cvq_forward(ctypes.byref(a), stream), 'vqhip_cvq_forward')
        all_reduce(packed[:a.exchange_floats])
        a.phases = STEP_AFTER_EXCHANGE
        check(L.vqhip_cvq_forward(ctypes.byref(a), stream), 'vqhip_cvq_forward')
    if use_host_word and prefetch and not capturing and not state.event_handle:
        state.event.record()
    return dict(idx=idx, hist=hist, xq=xq, prepared=ops.PreparedCodebook(image, w_in, K, D, m), z_ste=z_ste, mse=mse,
                cap_used=int(a.cap_used), exchange_floats=int(a.exchange_floats), x=x)


class VqkdStepState:
    """Arena of a VQKDCallback's fused forward."""

    def __init__(self) -> None:
        self.arena = _Arena()


@_on_tensor_device
def vqkd_forward(x: torch.Tensor, w_in: torch.Tensor, w_mid: torch.Tensor, w_out: torch.Tensor, metric, ema_decay: float,
                 state: VqkdStepState, *, exchange: bool, world: int, comm: Optional[int], all_reduce=None,
                 ordered: bool = False, tail: bool = True):
    """The VQ-KD training forward as one library call (two around a caller-issued collective, as ``cvq_forward``).
    Returns a dict: xn (F.normalize(x)), xq (F.normalize(xn)), idx, hist, prepared, z_ste, mse (fp32[4], [0] = the
    commitment loss with norm=True)."""
    ops._require_cuda(x, w_in, w_mid, w_out)
    x, dt = _latents(x)
    N, D = x.shape
    K = w_in.shape[0]
    m = METRICS[metric]
    L = _lib.lib()
    dev = x.device
    floats = ops.pack_floats(K, K, D)
    ws, packed = state.arena.get((N, K, D, dt), L.vqhip_vqkd_forward_ws_bytes(N, K, D), floats, dev)
    image = _bytes(L.vqhip_codebook_bytes(K, D), dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    hist = torch.empty(K, dtype=torch.int32, device=dev)
    xn = torch.empty(N, D, dtype=torch.float32, device=dev)
    xq = torch.empty(N, D, dtype=torch.float32, device=dev)
    z_ste = torch.empty(N, D, dtype=torch.float32, device=dev) if tail else None
    mse = torch.empty(4, dtype=torch.float32, device=dev) if tail else None
    a = _lib.VqkdForwardArgs()
    a.struct_bytes = ctypes.sizeof(_lib.VqkdForwardArgs)
    a.N, a.K, a.D, a.x_dtype, a.metric, a.world = N, K, D, dt, m, int(world)
    a.ema_decay = float(ema_decay)
    a.exchange, a.ordered, a.tail = int(bool(exchange)), int(bool(ordered)), int(bool(tail))
    a.x, a.w_in, a.w_mid, a.w_out = x.data_ptr(), w_in.data_ptr(), w_mid.data_ptr(), w_out.data_ptr()
    a.xn, a.xq = xn.data_ptr(), xq.data_ptr()
    a.comm = comm
    a.cb, a.cb_bytes = image.data_ptr(), image.numel()
    a.idx, a.hist = idx.data_ptr(), hist.data_ptr()
    a.packed, a.packed_floats = packed.data_ptr(), packed.numel()
    a.z_ste, a.mse = _p(z_ste), _p(mse)
    a.scratch16 = _mse_scratch(dev).data_ptr() if tail else None
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    stream = _stream()
    one_call = not exchange or comm is not None or (world == 1 and all_reduce is None)
    if one_call:
        a.phases = STEP_ALL
        check(L.vqhip_vqkd_forward(ctypes.byref(a), stream), 'vqhip_vqkd_forward')
    else:
        a.phases = STEP_BEFORE_EXCHANGE
        check(L.vqhip_vqkd_forward(ctypes.byref(a), stream), 'vqhip_vqkd_forward')
        all_reduce(packed[:floats])
        a.phases = STEP_AFTER_EXCHANGE
        check(L.vqhip_vqkd_forward(ctypes.byref(a), stream), 'vqhip_vqkd_forward')
    return dict(xn=xn, xq=xq, idx=idx, hist=hist, prepared=ops.PreparedCodebook(image, w_mid, K, D, m), z_ste=z_ste, mse=mse, x=x)


@_on_tensor_device
def vq_forward(x: torch.Tensor, w_in: torch.Tensor, w_out: Optional[torch.Tensor], metric, beta: float, *, normalize: bool,
               want_hist: bool, tail: bool = True):
    """The forward of a quantizer without an update callback — or with NormalizeCallback alone (``normalize``) — as one library
    call (include/vqhip.h: vqhip_vq_forward).  Returns a dict: xn (F.normalize(x), normalize only), idx, hist, xq (cosine),
    prepared, z_ste, mse, x (the latents as the library read them)."""
    ops._require_cuda(x, w_in)
    x, dt = _latents(x)
    N, D = x.shape
    K = w_in.shape[0]
    m = METRICS[metric]
    L = _lib.lib()
    dev = x.device
    cos = m in (_lib.METRIC_COS, _lib.METRIC_COS_BF16)
    ws = _bytes(L.vqhip_workspace_bytes(N, K, D), dev)
    image = _bytes(L.vqhip_codebook_bytes(K, D), dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    hist = torch.empty(K, dtype=torch.int32, device=dev) if want_hist else None
    xn = torch.empty(N, D, dtype=torch.float32, device=dev) if normalize else None
    xq = torch.empty(N, D, dtype=torch.float32, device=dev) if cos else None
    z_ste = torch.empty(N, D, dtype=torch.float32, device=dev) if tail else None
    mse = torch.empty(4, dtype=torch.float32, device=dev) if tail else None
    a = _lib.VqForwardArgs()
    a.struct_bytes = ctypes.sizeof(_lib.VqForwardArgs)
    a.N, a.K, a.D, a.x_dtype, a.metric, a.normalize = N, K, D, dt, m, int(bool(normalize))
    a.beta = float(beta)
    a.x, a.w_in, a.w_out, a.xn = x.data_ptr(), w_in.data_ptr(), _p(w_out), _p(xn)
    a.cb, a.cb_bytes = image.data_ptr(), image.numel()
    a.idx, a.hist, a.xq = idx.data_ptr(), _p(hist), _p(xq)
    a.z_ste, a.mse = _p(z_ste), _p(mse)
    a.scratch16 = _mse_scratch(dev).data_ptr() if tail else None
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    check(L.vqhip_vq_forward(ctypes.byref(a), _stream()), 'vqhip_vq_forward')
    codes = w_out if normalize else w_in
    return dict(xn=xn, idx=idx, hist=hist, xq=xq, prepared=ops.PreparedCodebook(image, codes, K, D, m), z_ste=z_ste, mse=mse, x=x)


@_on_tensor_device
def vqkd_backward(x: torch.Tensor, xn: torch.Tensor, w: torch.Tensor, idx: torch.Tensor, g_zste: Optional[torch.Tensor],
                  g_loss: Optional[torch.Tensor]) -> torch.Tensor:
    """grad_x of the VQ-KD tail (include/vqhip.h: vqhip_vqkd_backward) as fp32 [N, D]."""
    ops._require_cuda(x, xn, w, idx)
    x, dt = _latents(x)
    N, D = x.shape
    w = _codebook(w)
    gx = torch.empty(N, D, dtype=torch.float32, device=x.device)
    if g_zste is not None:
        g_zste = g_zste.float().contiguous()
    if g_loss is not None:
        g_loss = g_loss.detach().float().reshape(1).contiguous()
    check(_lib.lib().vqhip_vqkd_backward(x.data_ptr(), dt, xn.data_ptr(), w.data_ptr(), idx.data_ptr(), N, D, _p(g_zste), _p(g_loss),
                                         gx.data_ptr(), _stream()), 'vqhip_vqkd_backward')
    return gx
