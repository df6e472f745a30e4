"""Tensor-level front end of libvqhip: torch tensors in, torch tensors out, every byte of arithmetic in
the HIP library.  torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import METRIC_COS, METRIC_COS_BF16, METRIC_L2, check

# 'CosineBF16': cosine as the reference's GPU runs evaluate it under bf16 autocast (include/vqhip.h, VQHIP_METRIC_COS_BF16)
METRICS = {'L2': METRIC_L2, 'Cosine': METRIC_COS, 'CosineBF16': METRIC_COS_BF16,
           METRIC_L2: METRIC_L2, METRIC_COS: METRIC_COS, METRIC_COS_BF16: METRIC_COS_BF16}


_METRIC_NAMES = {METRIC_L2: 'L2', METRIC_COS: 'Cosine', METRIC_COS_BF16: 'CosineBF16'}


def metric_name(metric) -> str:
    """'L2' / 'Cosine' / 'CosineBF16' for a metric given by name or by its include/vqhip.h code."""
    return _METRIC_NAMES[METRICS[metric]]


def coarse_supported(D: int) -> bool:
    """True where the fp16 proposal image exists (D <= 1024, D % 8 == 0: every shipped config)."""
    return 1 <= D <= 1024 and D % 8 == 0


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of the current stream of the current device.  (The raw accessor: `torch.cuda.current_stream()`
    builds a Stream object per call — 11 us each, five calls per training step in the eager CVQ-VAE profile.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_tensor_device(fn):
    """Run the wrapped op with the first device tensor argument's device current: the launches go to that device's
    current stream, and the per-device kernel attributes (dynamic LDS size) are set for the right device.  A no-op when
    the tensors already live on the current device (the usual one-process-per-GPU case)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            t = a.weight if isinstance(a, PreparedCodebook) else a
            if isinstance(t, torch.Tensor) and t.is_cuda:
                if t.device.index != torch.cuda.current_device():
                    with torch.cuda.device(t.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


def _require_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VqhipError('vector_quantization_amd ops need tensors on an MI355X device (no CPU path)')


def _latents(x: torch.Tensor):
    """Dense row-major [N, D] view of the latents in a dtype the library reads (fp32 or bf16)."""
    if x.dim() != 2:
        raise ValueError(f'expected [N, D] latents, got {tuple(x.shape)}')
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.contiguous()
    if x.data_ptr() % 16:
        x = x.clone()
    return x, (_lib.DTYPE_F32 if x.dtype == torch.float32 else _lib.DTYPE_BF16)


def _codebook(e: torch.Tensor) -> torch.Tensor:
    if e.dim() != 2:
        raise ValueError(f'expected [K, D] codebook, got {tuple(e.shape)}')
    e = e.detach()
    if e.dtype != torch.float32:
        e = e.float()
    e = e.contiguous()
    if e.data_ptr() % 16:
        e = e.clone()
    return e


def _bytes(n: int, device) -> torch.Tensor:
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


@dataclass
class PreparedCodebook:
    """Device image produced by vqhip_codebook_prepare (fp16 MFMA fragments, |e|^2, error bounds)."""
    image: torch.Tensor
    weight: torch.Tensor          # the fp32 codebook the image was made from (kept alive for the re-rank)
    K: int
    D: int
    metric: int

    def exact_rows(self) -> Optional[torch.Tensor]:
        """Cosine images: the fp32 rows F.normalize(weight) the exact definition consumes, as a view into the image
        (bit-identical to ``normalize_rows(weight)``); None for L2 (the weight itself is the operand)."""
        if self.metric not in (METRIC_COS, METRIC_COS_BF16):
            return None
        off = _lib.lib().vqhip_codebook_exact_offset(self.K, self.D)
        return self.image[off:off + self.K * self.D * 4].view(torch.float32).view(self.K, self.D)


@_on_tensor_device
def prepare_codebook(e: torch.Tensor, metric='L2') -> PreparedCodebook:
    _require_cuda(e)
    e = _codebook(e)
    K, D = e.shape
    m = METRICS[metric]
    L = _lib.lib()
    image = _bytes(L.vqhip_codebook_bytes(K, D), e.device)
    check(L.vqhip_codebook_prepare(_ptr(e), K, D, m, _ptr(image), image.numel(), _stream()), 'vqhip_codebook_prepare')
    return PreparedCodebook(image, e, K, D, m)


@_on_tensor_device
def argmin(x: torch.Tensor, cb: PreparedCodebook, hist: Optional[torch.Tensor] = None,
           return_stats: bool = False):
    """idx[n] = argmin_k distance(x_n, e_k) — fused fp16 proposal + exact fp32 re-rank.

    For the cosine metric x must already be normalised (``normalize_rows``)."""
    _require_cuda(x)
    x, dt = _latents(x)
    N, D = x.shape
    if D != cb.D:
        raise ValueError(f'latent dim {D} != codebook dim {cb.D}')
    L = _lib.lib()
    idx = torch.empty(N, dtype=torch.int64, device=x.device)
    ws = _bytes(L.vqhip_workspace_bytes(N, cb.K, D), x.device)
    if hist is not None:
        assert hist.dtype == torch.int32 and hist.numel() == cb.K and hist.is_contiguous()
    check(L.vqhip_argmin(_ptr(x), dt, _ptr(cb.weight), _ptr(cb.image), cb.image.numel(), N, cb.K, D, cb.metric, _ptr(idx),
                         _ptr(hist), _ptr(ws), ws.numel(), _stream()), 'vqhip_argmin')
    if return_stats:
        st = torch.zeros(4, dtype=torch.int32, device=x.device)
        if N > 0:
            check(L.vqhip_argmin_stats(_ptr(ws), _ptr(st), _stream()), 'vqhip_argmin_stats')
        return idx, st
    return idx


@_on_tensor_device
def encode(x: torch.Tensor, e: torch.Tensor, metric='L2', hist: Optional[torch.Tensor] = None, zero_hist: bool = False):
    """``prepare_codebook`` + (cosine: ``normalize_rows(x)``) + ``argmin`` as ONE library call with two launches less:
    the training-time encode, where the codebook changes every step.  x are the latents as the quantizer receives them
    (not normalised).  Returns (idx, prepared codebook, xq) — xq = the normalised latents for cosine, None for L2;
    every value is that of the separate calls, bit for bit."""
    _require_cuda(x, e)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    if D != e.shape[1]:
        raise ValueError(f'latent dim {D} != codebook dim {e.shape[1]}')
    m = METRICS[metric]
    L = _lib.lib()
    image = _bytes(L.vqhip_codebook_bytes(K, D), e.device)
    idx = torch.empty(N, dtype=torch.int64, device=x.device)
    xq = torch.empty(N, D, dtype=torch.float32, device=x.device) if m in (METRIC_COS, METRIC_COS_BF16) else None
    ws = _bytes(L.vqhip_workspace_bytes(N, K, D), x.device)
    if hist is not None:
        assert hist.dtype == torch.int32 and hist.numel() == K and hist.is_contiguous()
    # zero_hist: `hist` may be uninitialised memory — the call's first launch zeroes it (no separate fill kernel)
    check(L.vqhip_encode_ex(_ptr(x), dt, _ptr(e), N, K, D, m, _ptr(image), image.numel(), _ptr(idx), _ptr(hist), _ptr(xq),
                            _ptr(ws), ws.numel(), 1 if (zero_hist and hist is not None) else 0, _stream()), 'vqhip_encode_ex')
    return idx, PreparedCodebook(image, e, K, D, m), xq


@_on_tensor_device
def encode_map(x_map: torch.Tensor, e: torch.Tensor, metric='L2', hist: Optional[torch.Tensor] = None, zero_hist: bool = False):
    """``encode`` for latents given as the NCHW-contiguous feature map [B, D, H, W]: 'b c h w -> (b h w) c' is folded into
    the call's first launch.  Returns (idx int64 [B*H*W], prepared codebook, xrows [B*H*W, D], xq) — xrows = the
    token-major copy of the latents in x's dtype, xq = the normalised fp32 rows (cosine; None for L2)."""
    _require_cuda(x_map, e)
    if x_map.dim() != 4 or not x_map.is_contiguous() or x_map.dtype not in (torch.float32, torch.bfloat16) or x_map.data_ptr() % 16:
        raise ValueError('encode_map expects a contiguous fp32 / bf16 [B, D, H, W] map')
    e = _codebook(e)
    B, D, H, W = x_map.shape
    K = e.shape[0]
    if D != e.shape[1]:
        raise ValueError(f'latent dim {D} != codebook dim {e.shape[1]}')
    m = METRICS[metric]
    L = _lib.lib()
    N = B * H * W
    dt = _lib.DTYPE_F32 if x_map.dtype == torch.float32 else _lib.DTYPE_BF16
    image = _bytes(L.vqhip_codebook_bytes(K, D), e.device)
    idx = torch.empty(N, dtype=torch.int64, device=x_map.device)
    cos = m in (METRIC_COS, METRIC_COS_BF16)
    xrows = torch.empty(N, D, dtype=x_map.dtype, device=x_map.device)
    xq = torch.empty(N, D, dtype=torch.float32, device=x_map.device) if cos else None
    ws = _bytes(L.vqhip_workspace_bytes(N, K, D), x_map.device)
    if hist is not None:
        assert hist.dtype == torch.int32 and hist.numel() == K and hist.is_contiguous()
    check(L.vqhip_encode_map(_ptr(x_map), dt, _ptr(e), B, H * W, K, D, m, _ptr(image), image.numel(), _ptr(idx), _ptr(hist),
                             _ptr(xrows), _ptr(xq), _ptr(ws), ws.numel(), 1 if (zero_hist and hist is not None) else 0, _stream()),
          'vqhip_encode_map')
    return idx, PreparedCodebook(image, e, K, D, m), xrows, xq


@_on_tensor_device
def gather_ste_map(x_rows: Optional[torch.Tensor], e: torch.Tensor, idx: torch.Tensor, B: int, H: int, W: int, beta: float = 0.0):
    """(out_map fp32 [B, D, H, W] NCHW-contiguous, mse fp32[4] or None): the straight-through output x + (e[idx] - x) — or,
    with ``x_rows`` None, the decoded rows e[idx] — written directly as the feature map ('(b h w) c -> b c h w' folded in)."""
    _require_cuda(e, idx)
    e = _codebook(e)
    D = e.shape[1]
    idx = idx.reshape(-1).contiguous()
    N = B * H * W
    assert idx.dtype == torch.int64 and idx.numel() == N and N > 0
    out = torch.empty(B, D, H, W, dtype=torch.float32, device=e.device)
    mse = None
    dt = _lib.DTYPE_F32
    scratch = None
    if x_rows is not None:
        x_rows, dt = _latents(x_rows)
        assert tuple(x_rows.shape) == (N, D)
        mse = torch.empty(4, dtype=torch.float32, device=e.device)
        scratch = _mse_scratch(e.device)
    check(_lib.lib().vqhip_gather_ste_map(_ptr(x_rows), dt, _ptr(e), _ptr(idx), B, H * W, D, _ptr(out), _ptr(mse), float(beta),
                                          _ptr(scratch), _stream()), 'vqhip_gather_ste_map')
    return out, mse


@_on_tensor_device
def argmin_exact(x: torch.Tensor, e: torch.Tensor, metric='L2', hist: Optional[torch.Tensor] = None,
                 return_min: bool = False):
    """Same contract as ``argmin`` evaluated entirely with fp32 MFMA (x, e normalised by the caller for cosine)."""
    _require_cuda(x, e)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    L = _lib.lib()
    idx = torch.empty(N, dtype=torch.int64, device=x.device)
    dmin = torch.empty(N, dtype=torch.float32, device=x.device) if return_min else None
    ws = _bytes(L.vqhip_workspace_bytes(N, K, D), x.device)
    check(L.vqhip_argmin_exact(_ptr(x), dt, _ptr(e), N, K, D, METRICS[metric], _ptr(idx), _ptr(dmin), _ptr(hist),
                               _ptr(ws), ws.numel(), _stream()), 'vqhip_argmin_exact')
    return (idx, dmin) if return_min else idx


@_on_tensor_device
def distance(x: torch.Tensor, e: torch.Tensor, metric='L2') -> torch.Tensor:
    """Materialised d[N, K] (memo['distance']); x, e normalised by the caller for cosine."""
    _require_cuda(x, e)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    L = _lib.lib()
    d = torch.empty(N, K, dtype=torch.float32, device=x.device)
    ws = _bytes(L.vqhip_workspace_bytes(N, K, D), x.device)
    check(L.vqhip_distance(_ptr(x), dt, _ptr(e), N, K, D, METRICS[metric], _ptr(d), _ptr(ws), ws.numel(), _stream()),
          'vqhip_distance')
    return d


@_on_tensor_device
def col_argmin(x: torch.Tensor, e: torch.Tensor, metric='L2') -> torch.Tensor:
    """NearestAnchor indices: for every code the nearest token (lowest token on ties)."""
    _require_cuda(x, e)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    L = _lib.lib()
    out = torch.empty(K, dtype=torch.int64, device=x.device)
    ws = _bytes(L.vqhip_col_workspace_bytes(N, K, D), x.device)
    check(L.vqhip_col_argmin(_ptr(x), dt, _ptr(e), N, K, D, METRICS[metric], _ptr(out), _ptr(ws), ws.numel(), _stream()),
          'vqhip_col_argmin')
    return out


@_on_tensor_device
def row_sqnorm(v: torch.Tensor) -> torch.Tensor:
    _require_cuda(v)
    v, dt = _latents(v)
    out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    check(_lib.lib().vqhip_row_sqnorm(_ptr(v), dt, v.shape[0], v.shape[1], _ptr(out), _stream()), 'vqhip_row_sqnorm')
    return out


@_on_tensor_device
def normalize_rows(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(v, dim=1) as fp32."""
    _require_cuda(v)
    v, dt = _latents(v)
    out = torch.empty(v.shape, dtype=torch.float32, device=v.device)
    check(_lib.lib().vqhip_normalize_rows(_ptr(v), dt, v.shape[0], v.shape[1], eps, _ptr(out), _stream()),
          'vqhip_normalize_rows')
    return out


@_on_tensor_device
def gather_ste_loss(x: torch.Tensor, e: torch.Tensor, idx: torch.Tensor, need_z: bool = True,
                    need_ste: bool = True, need_sse: bool = True):
    """(z = e[idx], z_ste = x + (z - x), sse = sum (z-x)^2 as a float64[1] tensor); unwanted outputs are None."""
    _require_cuda(x, e, idx)
    x, dt = _latents(x)
    e = _codebook(e)
    idx = idx.reshape(-1).contiguous()
    assert idx.dtype == torch.int64 and idx.numel() == x.shape[0]
    N, D = x.shape
    z = torch.empty(N, D, dtype=torch.float32, device=x.device) if need_z else None
    zs = torch.empty(N, D, dtype=torch.float32, device=x.device) if need_ste else None
    sse = torch.zeros(1, dtype=torch.float64, device=x.device) if need_sse else None
    check(_lib.lib().vqhip_gather_ste_loss(_ptr(x), dt, _ptr(e), _ptr(idx), N, D, _ptr(z), _ptr(zs), _ptr(sse),
                                           _stream()), 'vqhip_gather_ste_loss')
    return z, zs, sse


_MSE_SCRATCH: dict = {}      # (device index, stream) -> 16 zeroed bytes the kernel hands back zeroed


_MSE_SCRATCH_OWNED: list = []    # innermost `owned_mse_scratch` buffer, if any


class owned_mse_scratch:
    """While active, ``gather_ste_mse`` uses ``buf`` (16 zeroed device bytes the caller owns) instead of the per-stream
    cache.  GraphedQuantizer captures with its own buffer: two captured graphs never share a scratch (they could replay
    concurrently on different streams), and no zero-fill is captured into the graph."""

    def __init__(self, buf: torch.Tensor) -> None:
        assert buf.is_cuda and buf.numel() * buf.element_size() >= 16
        self.buf = buf

    def __enter__(self):
        _MSE_SCRATCH_OWNED.append(self.buf)
        return self.buf

    def __exit__(self, *exc):
        _MSE_SCRATCH_OWNED.pop()
        return False


def _mse_scratch(device: torch.device) -> torch.Tensor:
    if _MSE_SCRATCH_OWNED and _MSE_SCRATCH_OWNED[-1].device == device:
        return _MSE_SCRATCH_OWNED[-1]
    key = (device.index, _raw_stream(device.index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
    buf = _MSE_SCRATCH.get(key)
    if buf is None:
        buf = _MSE_SCRATCH[key] = torch.zeros(16, dtype=torch.uint8, device=device)
    return buf


@_on_tensor_device
def gather_ste_mse(x: torch.Tensor, e: torch.Tensor, idx: torch.Tensor, need_z: bool = False, need_ste: bool = True,
                   beta: float = 0.0):
    """(z or None, z_ste or None, mse fp32[4]) with mse[0] = mse[1] = mean((e[idx] - x)^2) and mse[2] = mse[0] + beta*mse[1]
    (VQGANLoss): `gather_ste_loss` with the mean finished inside the kernel (no zero-fill, division, cast, scale and add
    kernels around it)."""
    _require_cuda(x, e, idx)
    x, dt = _latents(x)
    e = _codebook(e)
    idx = idx.reshape(-1).contiguous()
    assert idx.dtype == torch.int64 and idx.numel() == x.shape[0] and x.shape[0] > 0
    N, D = x.shape
    z = torch.empty(N, D, dtype=torch.float32, device=x.device) if need_z else None
    zs = torch.empty(N, D, dtype=torch.float32, device=x.device) if need_ste else None
    mse = torch.empty(4, dtype=torch.float32, device=x.device)
    check(_lib.lib().vqhip_gather_ste_mse(_ptr(x), dt, _ptr(e), _ptr(idx), N, D, _ptr(z), _ptr(zs), _ptr(mse), float(beta),
                                          _ptr(_mse_scratch(x.device)), _stream()), 'vqhip_gather_ste_mse')
    return z, zs, mse


@_on_tensor_device
def hist(idx: torch.Tensor, K: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_cuda(idx)
    idx = idx.reshape(-1).contiguous()
    assert idx.dtype == torch.int64
    if out is None:
        out = torch.zeros(K, dtype=torch.int32, device=idx.device)
    check(_lib.lib().vqhip_hist(_ptr(idx), idx.numel(), K, _ptr(out), _stream()), 'vqhip_hist')
    return out


ORDERED_MAX_K = 32768     # the ordered (deterministic) route keeps one code histogram per 1024-token chunk in LDS


def use_ordered(K: int, D: int, ordered: Optional[bool] = None, N: Optional[int] = None, backward: bool = False) -> bool:
    """Policy of the codebook-side sums (k-means centroid sums, codebook gradient).  The ordered route — tokens sorted
    by code, sums in a fixed order, bit-reproducible — is taken when asked for explicitly, under
    ``torch.use_deterministic_algorithms(True)``, and by default where it is also the faster one on MI355X (measured,
    tools/bench_ordered.py): centroid sums from N >= 32768; the fused backward (``backward=True``) from N >= 131072, or
    from N >= 32768 on contended small codebooks (K <= 4096).
    Otherwise fp32 atomics.  It needs K <= 32768 and D % 4 == 0.  ``VQHIP_ORDERED=0/1`` overrides the default."""
    ok = K <= ORDERED_MAX_K and D % 4 == 0
    if ordered is None:
        if torch.are_deterministic_algorithms_enabled():
            if not ok:
                raise RuntimeError(f'vector_quantization_amd: no deterministic codebook-side sum for K={K}, D={D} '
                                   f'(needs K <= {ORDERED_MAX_K} and D % 4 == 0)')
            return True
        env = os.environ.get('VQHIP_ORDERED')
        if env is not None:
            return env != '0' and ok
        if not ok or N is None:
            return False
        if backward:
            return N >= 131072 or (K <= 4096 and N >= 32768)
        return N >= 32768
    if ordered and not ok:
        raise ValueError(f'ordered sums need K <= {ORDERED_MAX_K} and D % 4 == 0 (got K={K}, D={D})')
    return bool(ordered)


@_on_tensor_device
def token_order(idx: torch.Tensor, K: int):
    """Stable counting sort of the token ids by code: (counts int32[K], offsets int32[K+1], order int32[N])."""
    _require_cuda(idx)
    idx = idx.reshape(-1).contiguous()
    assert idx.dtype == torch.int64
    N = idx.numel()
    L = _lib.lib()
    counts = torch.empty(K, dtype=torch.int32, device=idx.device)
    offsets = torch.empty(K + 1, dtype=torch.int32, device=idx.device)
    order = torch.empty(max(N, 1), dtype=torch.int32, device=idx.device)
    ws = _bytes(L.vqhip_order_workspace_bytes(N, K), idx.device)
    check(L.vqhip_token_order(_ptr(idx), N, K, _ptr(counts), _ptr(offsets), _ptr(order), _ptr(ws), ws.numel(), _stream()),
          'vqhip_token_order')
    return counts, offsets, order[:N]


@_on_tensor_device
def segsum_rows(src: torch.Tensor, idx: torch.Tensor, offsets: torch.Tensor, order: torch.Tensor, K: int) -> torch.Tensor:
    """out[k] = sum of src[order[p]] for p in [offsets[k], offsets[k+1]) in the fixed blocked order of the ordered route
    (fp32; see include/vqhip.h)."""
    _require_cuda(src, idx, offsets, order)
    src = src.float().contiguous()
    N, D = src.shape
    L = _lib.lib()
    out = torch.empty(K, D, dtype=torch.float32, device=src.device)
    ws = _bytes(L.vqhip_segsum_workspace_bytes(N, D), src.device)
    check(L.vqhip_segsum_rows(_ptr(src), _ptr(idx), _ptr(order), _ptr(offsets), N, K, D, _ptr(out), _ptr(ws), ws.numel(),
                              _stream()), 'vqhip_segsum_rows')
    return out


@_on_tensor_device
def scatter_add_rows(src: torch.Tensor, idx: torch.Tensor, K: int, out: Optional[torch.Tensor] = None,
                     ordered: Optional[bool] = None) -> torch.Tensor:
    """out[idx[n]] += src[n] (centroid sums / dense embedding backward).  Ordered route (see ``use_ordered``) when no
    accumulator is passed in; fp32 atomics otherwise."""
    _require_cuda(src, idx)
    src = src.float().contiguous()
    idx = idx.reshape(-1).contiguous()
    if out is None and use_ordered(K, src.shape[1], ordered, src.shape[0]):
        _, offsets, order = token_order(idx, K)
        return segsum_rows(src, idx, offsets, order, K)
    if out is None:
        out = torch.zeros(K, src.shape[1], dtype=torch.float32, device=src.device)
    check(_lib.lib().vqhip_scatter_add_rows(_ptr(src), _ptr(idx), src.shape[0], K, src.shape[1], _ptr(out), _stream()),
          'vqhip_scatter_add_rows')
    return out


@_on_tensor_device
def gather_rows(x: torch.Tensor, row_idx: torch.Tensor) -> torch.Tensor:
    _require_cuda(x, row_idx)
    x, dt = _latents(x)
    row_idx = row_idx.reshape(-1).contiguous()
    out = torch.empty(row_idx.numel(), x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.lib().vqhip_gather_rows(_ptr(x), dt, _ptr(row_idx), row_idx.numel(), x.shape[1], _ptr(out), _stream()),
          'vqhip_gather_rows')
    return out


@_on_tensor_device
def vqkd_update_(w: torch.Tensor, hist64: torch.Tensor, sums: torch.Tensor, decay: float, mode: str = 'full') -> torch.Tensor:
    """In-place VQ-KD codebook update on a contiguous fp32 [K, D] tensor (mode 'centroid': k-means centroids only)."""
    _require_cuda(w, hist64, sums)
    assert w.dtype == torch.float32 and w.is_contiguous() and hist64.dtype == torch.int64
    K, D = w.shape
    hist64, sums = hist64.contiguous(), sums.contiguous()      # named: they must outlive the enqueue
    check(_lib.lib().vqhip_vqkd_update(_ptr(w), _ptr(hist64), _ptr(sums), K, D, decay,
                                       1 if mode == 'centroid' else 0, _stream()), 'vqhip_vqkd_update')
    return w


@_on_tensor_device
def cvq_update_(w: torch.Tensor, p: torch.Tensor, hist64: Optional[torch.Tensor], numel, anchors: Optional[torch.Tensor],
                ema_decay: float, eps: float, stage: int = 3) -> None:
    """In-place CVQ-VAE update; stage 1 = probability only, 2 = codebook only (from the current p), 3 = both.
    ``numel`` is an int or a device int64 scalar tensor."""
    _require_cuda(w, p)
    assert w.dtype == torch.float32 and w.is_contiguous() and p.dtype == torch.float32 and p.is_contiguous()
    K, D = w.shape
    numel_dev = numel if isinstance(numel, torch.Tensor) else None
    if numel_dev is not None:
        numel_dev = numel_dev.to(torch.int64).reshape(1)
        assert numel_dev.is_cuda
    hist64 = None if hist64 is None else hist64.contiguous()
    anchors = None if anchors is None else anchors.contiguous()
    check(_lib.lib().vqhip_cvq_update(_ptr(w), _ptr(p), _ptr(hist64), 0 if (numel_dev is not None or numel is None)
                                      else int(numel), _ptr(numel_dev), _ptr(anchors), K, D, ema_decay, eps, stage,
                                      _stream()), 'vqhip_cvq_update')


@_on_tensor_device
def cvq_step(w_in: torch.Tensor, w_out: torch.Tensor, p_in: torch.Tensor, p_out: torch.Tensor, hist32: torch.Tensor,
             numel: int, x: torch.Tensor, col_idx: torch.Tensor, ema_decay: float, eps: float) -> None:
    """The one-rank CVQ-VAE update (probability EMA, decay, NearestAnchor gather, blend) in one launch; ``w_out`` /
    ``p_out`` may be ``w_in`` / ``p_in`` themselves."""
    _require_cuda(w_in, w_out, p_in, p_out, hist32, x, col_idx)
    x, dt = _latents(x)
    K, D = w_in.shape
    for t in (w_in, w_out):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (K, D)
    for t in (p_in, p_out):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == K
    assert hist32.dtype == torch.int32 and hist32.is_contiguous() and hist32.numel() == K
    col_idx = col_idx.to(torch.int64).contiguous()
    assert col_idx.numel() == K and x.shape[1] == D
    check(_lib.lib().vqhip_cvq_step(_ptr(w_in), _ptr(w_out), _ptr(p_in), _ptr(p_out), _ptr(hist32), int(numel), _ptr(x), dt,
                                    _ptr(col_idx), K, D, ema_decay, eps, _stream()), 'vqhip_cvq_step')


@_on_tensor_device
def cvq_decay(p: torch.Tensor, K: int, ema_decay: float, eps: float) -> torch.Tensor:
    """decay_k = 1 - exp(-p_k*K*10/(1-ema_decay) - eps) with the update kernel's own expression (fp32 [K])."""
    _require_cuda(p)
    assert p.dtype == torch.float32 and p.is_contiguous()
    out = torch.empty(K, dtype=torch.float32, device=p.device)
    check(_lib.lib().vqhip_cvq_decay(_ptr(p), K, ema_decay, eps, _ptr(out), _stream()), 'vqhip_cvq_decay')
    return out


@_on_tensor_device
def cvq_update_rows_(w: torch.Tensor, p: torch.Tensor, rows: torch.Tensor, anchors_sub: torch.Tensor, ema_decay: float,
                     eps: float) -> None:
    """In place: w[rows[i]] = w[rows[i]]*decay + anchors_sub[i]*(1-decay) — stage 2 of the CVQ-VAE update for the
    listed codes only (the others have decay == 1 and would keep their weight anyway)."""
    _require_cuda(w, p, rows, anchors_sub)
    assert w.dtype == torch.float32 and w.is_contiguous() and p.dtype == torch.float32 and p.is_contiguous()
    rows = rows.to(torch.int64).contiguous()
    anchors_sub = anchors_sub.float().contiguous()
    K, D = w.shape
    assert anchors_sub.shape == (rows.numel(), D)
    check(_lib.lib().vqhip_cvq_update_rows(_ptr(w), _ptr(p), _ptr(rows), _ptr(anchors_sub), rows.numel(), K, D, ema_decay,
                                           eps, _stream()), 'vqhip_cvq_update_rows')


# ---- CVQ-VAE with anchors for the codes that can need one, and the packed exchange (include/vqhip.h) ----------------

@_on_tensor_device
def cvq_rows(p: torch.Tensor, K: int, ema_decay: float, eps: float, out=None):
    """(rows int32[K], slot int32[K], count int32[1]) — the codes whose decay can come out below 1 in the step that starts
    from the probabilities ``p`` (ascending; slot[k] = position in rows or -1).  ``out``: reuse those three tensors."""
    _require_cuda(p)
    assert p.dtype == torch.float32 and p.is_contiguous() and p.numel() == K
    if out is None:
        rows = torch.empty(K, dtype=torch.int32, device=p.device)
        slot = torch.empty(K, dtype=torch.int32, device=p.device)
        count = torch.empty(1, dtype=torch.int32, device=p.device)
    else:
        rows, slot, count = out
    check(_lib.lib().vqhip_cvq_rows(_ptr(p), K, ema_decay, eps, _ptr(rows), _ptr(slot), _ptr(count), _stream()), 'vqhip_cvq_rows')
    return rows, slot, count


@_on_tensor_device
def col_argmin_rows(x: torch.Tensor, e: torch.Tensor, rows: torch.Tensor, count: torch.Tensor, cap: int, metric='L2') -> torch.Tensor:
    """NearestAnchor indices of the listed codes only: out[i] = nearest latent of code rows[i] for i < count (int64 [cap];
    entries past the count are unspecified).  ``cap`` >= the count sizes the launches."""
    _require_cuda(x, e, rows, count)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    assert rows.dtype == torch.int32 and count.dtype == torch.int32 and 0 <= cap <= K
    L = _lib.lib()
    out = torch.empty(max(cap, 1), dtype=torch.int64, device=x.device)
    if cap > 0:
        ws = _bytes(L.vqhip_col_rows_workspace_bytes(N, cap, D), x.device)
        check(L.vqhip_col_argmin_rows(_ptr(x), dt, _ptr(e), _ptr(rows), _ptr(count), cap, N, K, D, METRICS[metric], _ptr(out),
                                      _ptr(ws), ws.numel(), _stream()), 'vqhip_col_argmin_rows')
    return out[:cap]


def pack_floats(K: int, M: int, D: int) -> int:
    return int(_lib.lib().vqhip_pack_floats(K, M, D))


@_on_tensor_device
def pack_counts(hist: torch.Tensor, numel: int, packed: torch.Tensor) -> torch.Tensor:
    """Header of the packed exchange buffer from an int32 / int64 histogram and this rank's token count."""
    _require_cuda(hist, packed)
    K = hist.numel()
    assert hist.dtype in (torch.int32, torch.int64) and hist.is_contiguous()
    assert packed.dtype == torch.float32 and packed.is_contiguous() and packed.numel() >= 2 * K + 4
    check(_lib.lib().vqhip_pack_counts(_ptr(hist), 1 if hist.dtype == torch.int64 else 0, int(numel), K, _ptr(packed), _stream()),
          'vqhip_pack_counts')
    return packed


@_on_tensor_device
def unpack_counts(packed: torch.Tensor, K: int) -> torch.Tensor:
    """int64 [K + 1] = code counts ‖ token count of an (all-reduced) packed buffer."""
    _require_cuda(packed)
    assert packed.dtype == torch.float32 and packed.is_contiguous()
    out = torch.empty(K + 1, dtype=torch.int64, device=packed.device)
    check(_lib.lib().vqhip_unpack_counts(_ptr(packed), K, _ptr(out), _stream()), 'vqhip_unpack_counts')
    return out


@_on_tensor_device
def cvq_pack(hist32: torch.Tensor, numel: int, x: torch.Tensor, col_idx: torch.Tensor, count: torch.Tensor, cap: int, K: int) -> torch.Tensor:
    """This rank's packed buffer for the CVQ-VAE exchange: fp32 [2K + 4 + cap*D]."""
    _require_cuda(hist32, x)
    x, dt = _latents(x)
    D = x.shape[1]
    assert hist32.dtype == torch.int32 and hist32.is_contiguous() and hist32.numel() == K
    packed = torch.empty(pack_floats(K, cap, D), dtype=torch.float32, device=x.device)
    check(_lib.lib().vqhip_cvq_pack(_ptr(hist32), int(numel), _ptr(x), dt, _ptr(col_idx) if cap else None, _ptr(count) if cap else None,
                                    cap, K, D, _ptr(packed), _stream()), 'vqhip_cvq_pack')
    return packed


@_on_tensor_device
def cvq_apply(w_in: torch.Tensor, w_out: torch.Tensor, p_in: torch.Tensor, p_out: torch.Tensor, slot: torch.Tensor,
              ema_decay: float, eps: float, hist32: Optional[torch.Tensor] = None, numel: int = 0,
              x: Optional[torch.Tensor] = None, col_idx: Optional[torch.Tensor] = None,
              packed: Optional[torch.Tensor] = None, world: int = 1, cap: Optional[int] = None) -> None:
    """The CVQ-VAE update with anchors for the listed codes (``slot``): from the all-reduced ``packed`` buffer of ``world``
    ranks, or (one rank) from ``hist32`` / ``numel`` / ``x`` / ``col_idx``.  ``w_out`` / ``p_out`` may alias the inputs.
    ``cap``: the capacity the column pass / the pack were sized for (default: what the buffers handed in hold)."""
    _require_cuda(w_in, w_out, p_in, p_out, slot)
    K, D = w_in.shape
    for t in (w_in, w_out):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (K, D)
    for t in (p_in, p_out):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == K
    assert slot.dtype == torch.int32 and slot.numel() == K
    dt = _lib.DTYPE_F32
    if packed is None:
        assert hist32 is not None and hist32.dtype == torch.int32 and hist32.is_contiguous() and numel > 0
        if x is not None:
            x, dt = _latents(x)
            assert x.shape[1] == D
    else:
        assert packed.dtype == torch.float32 and packed.is_contiguous()
    if cap is None:
        cap = (packed.numel() - (2 * K + 4)) // D if packed is not None else (col_idx.numel() if col_idx is not None else 0)
    cap = max(0, min(int(cap), K))
    check(_lib.lib().vqhip_cvq_apply(_ptr(w_in), _ptr(w_out), _ptr(p_in), _ptr(p_out), _ptr(hist32), int(numel), _ptr(x), dt,
                                     _ptr(col_idx), _ptr(packed), int(world), _ptr(slot), cap, K, D, ema_decay, eps, _stream()),
          'vqhip_cvq_apply')


SYNC_MAX_ROWS = 1 << 24


@_on_tensor_device
def cvq_col_keys(x: torch.Tensor, e: torch.Tensor, rows: torch.Tensor, count: torch.Tensor, cap: int, col_idx: torch.Tensor,
                 metric, rank: int) -> torch.Tensor:
    """NearestAnchor(sync=True) across ranks (include/vqhip.h: vqhip_cvq_col_keys): int64 [cap] keys of this rank's local column
    winners — (distance, rank, row) in torch.argmin's order, MIN-all-reducible as signed integers.  ``x`` / ``e`` are the operands
    ``col_argmin_rows`` was given."""
    _require_cuda(x, e, rows, count, col_idx)
    x, dt = _latents(x)
    e = _codebook(e)
    N, D = x.shape
    K = e.shape[0]
    keys = torch.empty(max(cap, 1), dtype=torch.int64, device=x.device)
    if cap > 0:
        check(_lib.lib().vqhip_cvq_col_keys(_ptr(x), dt, _ptr(e), _ptr(rows), _ptr(count), cap, _ptr(col_idx), N, K, D, METRICS[metric],
                                            int(rank), _ptr(keys), _stream()), 'vqhip_cvq_col_keys')
    return keys[:cap]


@_on_tensor_device
def cvq_pack_sync(hist32: torch.Tensor, numel: int, x: torch.Tensor, keys: torch.Tensor, count: torch.Tensor, cap: int, K: int,
                  rank: int) -> torch.Tensor:
    """``cvq_pack`` behind the MIN all-reduce of the keys: payload row i = x[row] on the rank the reduced key names, -0.0 elsewhere."""
    _require_cuda(hist32, x)
    x, dt = _latents(x)
    D = x.shape[1]
    assert hist32.dtype == torch.int32 and hist32.is_contiguous() and hist32.numel() == K
    packed = torch.empty(pack_floats(K, cap, D), dtype=torch.float32, device=x.device)
    check(_lib.lib().vqhip_cvq_pack_sync(_ptr(hist32), int(numel), _ptr(x), dt, _ptr(keys) if cap else None, _ptr(count) if cap else None,
                                         cap, int(rank), K, D, _ptr(packed), _stream()), 'vqhip_cvq_pack_sync')
    return packed


def _any(t: torch.Tensor):
    """Flat contiguous fp32/bf16 view + dtype code for the elementwise kernels."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    t = t.contiguous()
    return t, (_lib.DTYPE_F32 if t.dtype == torch.float32 else _lib.DTYPE_BF16)


@_on_tensor_device
def sse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sum((a-b)^2) as a float64[1] device tensor."""
    _require_cuda(a, b)
    a, da = _any(a)
    b, db = _any(b)
    assert a.numel() == b.numel()
    out = torch.zeros(1, dtype=torch.float64, device=a.device)
    check(_lib.lib().vqhip_diff(_ptr(a), da, _ptr(b), db, a.numel(), 1.0, None, None, _ptr(out), _stream()),
          'vqhip_diff')
    return out


@_on_tensor_device
def diff_scale(a: torch.Tensor, b: torch.Tensor, scale: float, scale_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(a - b) * scale [* scale_dev] as fp32 (the MSE backward; scale_dev = upstream scalar gradient on the device)."""
    _require_cuda(a, b)
    shape = a.shape
    a, da = _any(a)
    b, db = _any(b)
    out = torch.empty(a.numel(), dtype=torch.float32, device=a.device)
    if scale_dev is not None:
        scale_dev = scale_dev.detach().float().reshape(1).contiguous()
    check(_lib.lib().vqhip_diff(_ptr(a), da, _ptr(b), db, a.numel(), float(scale), _ptr(scale_dev), _ptr(out), None,
                                _stream()), 'vqhip_diff')
    return out.view(shape)


@_on_tensor_device
def ste(x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """x + (z - x) as fp32."""
    _require_cuda(x, z)
    shape = z.shape
    x, dx = _any(x)
    z = z.float().contiguous()
    out = torch.empty(z.numel(), dtype=torch.float32, device=z.device)
    check(_lib.lib().vqhip_ste(_ptr(x), dx, _ptr(z), z.numel(), _ptr(out), _stream()), 'vqhip_ste')
    return out.view(shape)


@_on_tensor_device
def normalize_rows_bwd(v: torch.Tensor, g: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    _require_cuda(v, g)
    v, dt = _latents(v)
    g = g.float().contiguous()
    out = torch.empty(v.shape, dtype=torch.float32, device=v.device)
    check(_lib.lib().vqhip_normalize_rows_bwd(_ptr(v), dt, _ptr(g), v.shape[0], v.shape[1], eps, _ptr(out), _stream()),
          'vqhip_normalize_rows_bwd')
    return out


@_on_tensor_device
def vq_backward(x: torch.Tensor, e: torch.Tensor, idx: torch.Tensor, g_zste: Optional[torch.Tensor],
                g_cb: Optional[torch.Tensor], g_cm: Optional[torch.Tensor], need_x: bool, need_w: bool,
                ordered: Optional[bool] = None, g_comb: Optional[torch.Tensor] = None, beta: float = 0.0):
    """Fused backward of the quantizer forward; returns (grad_x fp32 or None, grad_w fp32 [K, D] or None).  The
    codebook gradient is summed code by code in token order (``use_ordered``) or with fp32 atomics."""
    _require_cuda(x, e, idx)
    x, dt = _latents(x)
    e = _codebook(e)
    idx = idx.reshape(-1).contiguous()
    N, D = x.shape
    K = e.shape[0]
    if not need_x and not need_w:
        return None, None
    L = _lib.lib()
    gx = torch.empty(N, D, dtype=torch.float32, device=x.device) if need_x else None
    if g_zste is not None:
        g_zste = g_zste.float().contiguous()
    scal = [None if g is None else g.detach().float().reshape(1).contiguous() for g in (g_cb, g_cm, g_comb)]
    ordered_w = need_w and use_ordered(K, D, ordered, N, backward=True)
    if ordered_w and scal[2] is not None:        # the ordered kernel takes one scalar: fold the combined gradient in here
        scal[0] = scal[2] if scal[0] is None else scal[0] + scal[2]
    gw = None
    if need_w:
        gw = torch.empty(e.shape, dtype=torch.float32, device=x.device) if ordered_w else \
            torch.zeros(e.shape, dtype=torch.float32, device=x.device)
    if need_x or (need_w and not ordered_w):
        check(L.vqhip_vq_backward_ex(_ptr(x), dt, _ptr(e), _ptr(idx), N, D, _ptr(g_zste), _ptr(scal[0]), _ptr(scal[1]),
                                     _ptr(scal[2]), float(beta), _ptr(gx), None if ordered_w else _ptr(gw), _stream()),
              'vqhip_vq_backward_ex')
    if ordered_w:
        _, offsets, order = token_order(idx, K)
        ws = _bytes(L.vqhip_segsum_workspace_bytes(N, D), x.device)
        check(L.vqhip_vq_backward_w_ordered(_ptr(x), dt, _ptr(e), _ptr(idx), _ptr(order), _ptr(offsets), N, K, D,
                                            _ptr(scal[0]), _ptr(gw), _ptr(ws), ws.numel(), _stream()), 'vqhip_vq_backward_w_ordered')
    return gx, gw


def backward_map_supported(D: int, hw: int) -> bool:
    """Shapes ``vq_backward_map`` takes (whole 1 KiB channel rows per wave-instruction): HW % 256 == 0, D % 32 == 0."""
    return hw % 256 == 0 and D % 32 == 0


@_on_tensor_device
def vq_backward_map(x_rows: torch.Tensor, e: torch.Tensor, idx: torch.Tensor, shape, g_map: Optional[torch.Tensor],
                    g_cm: Optional[torch.Tensor], g_comb: Optional[torch.Tensor], beta: float, out_dtype: torch.dtype) -> torch.Tensor:
    """Gradient of the latents of a quantizer call on the NCHW map [B, D, H, W], written as that map in ``out_dtype`` (fp32 or
    bf16) from the upstream gradient ``g_map`` of the straight-through output given as the map too (include/vqhip.h:
    vqhip_vq_backward_map — no transposes, no cast)."""
    _require_cuda(x_rows, e, idx)
    b, d, h, w = shape
    x_rows, dt = _latents(x_rows)
    e = _codebook(e)
    idx = idx.reshape(-1).contiguous()
    assert tuple(x_rows.shape) == (b * h * w, d) and out_dtype in (torch.float32, torch.bfloat16)
    if g_map is not None:
        g_map = g_map.float().contiguous()
        assert tuple(g_map.shape) == (b, d, h, w)
    scal = [None if g is None else g.detach().float().reshape(1).contiguous() for g in (g_cm, g_comb)]
    out = torch.empty(b, d, h, w, dtype=out_dtype, device=x_rows.device)
    check(_lib.lib().vqhip_vq_backward_map(_ptr(x_rows), dt, _ptr(e), _ptr(idx), b, h * w, d, _ptr(g_map), _ptr(scal[0]), _ptr(scal[1]),
                                           float(beta), _ptr(out), _lib.DTYPE_F32 if out_dtype == torch.float32 else _lib.DTYPE_BF16,
                                           _stream()), 'vqhip_vq_backward_map')
    return out


@_on_tensor_device
def transpose_last2(t: torch.Tensor) -> torch.Tensor:
    """[B, R, C] -> [B, C, R] for fp32 / bf16 / fp16 tensors (the BCHW <-> (BHW)C rearrangement)."""
    _require_cuda(t)
    assert t.dim() == 3 and t.element_size() in (2, 4)
    t = t.contiguous()
    B, R, C = t.shape
    out = torch.empty(B, C, R, dtype=t.dtype, device=t.device)
    check(_lib.lib().vqhip_transpose(_ptr(t), _ptr(out), t.element_size(), B, R, C, _stream()), 'vqhip_transpose')
    return out


@_on_tensor_device
def codebook_metrics(counts: torch.Tensor) -> torch.Tensor:
    """float64[2] device tensor: (usage = nonzero/K, entropy in nats) of an int64 count vector."""
    _require_cuda(counts)
    counts = counts.to(torch.int64).contiguous()
    out = torch.empty(2, dtype=torch.float64, device=counts.device)
    check(_lib.lib().vqhip_codebook_metrics(_ptr(counts), counts.numel(), _ptr(out), _stream()), 'vqhip_codebook_metrics')
    return out


@_on_tensor_device
def debug_proposal_scores(x: torch.Tensor, cb: PreparedCodebook):
    """(scores[N, K], margin[N], scale) of the fp16 proposal pass — verification aid for the error-bound tests."""
    _require_cuda(x)
    x, dt = _latents(x)
    N, D = x.shape
    L = _lib.lib()
    scores = torch.empty(N, cb.K, dtype=torch.float32, device=x.device)
    margin = torch.empty(N, dtype=torch.float32, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = _bytes(L.vqhip_workspace_bytes(N, cb.K, D), x.device)
    check(L.vqhip_debug_proposal_scores(_ptr(x), dt, _ptr(cb.image), N, cb.K, D, cb.metric, _ptr(scores), _ptr(margin),
                                        _ptr(scale), _ptr(ws), ws.numel(), _stream()), 'vqhip_debug_proposal_scores')
    return scores, margin, scale
