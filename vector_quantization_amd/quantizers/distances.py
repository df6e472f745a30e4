"""Distances — mirror of vq/algorithms/vq/distances.py:19-46.

``forward(x, e)`` materialises d[N, K] like the reference, WITH autograd (forward = the HIP fp32-MFMA distance kernel,
backward = two plain library GEMMs; used only by consumers that need the matrix); ``argmin(x, e)`` is the fused hot
path that never forms it."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch
from torch import nn
from torch.autograd import Function

from .. import functional as VF
from .. import ops
from ..registries import VQITQuantizerDistanceRegistry


class _L2Matrix(Function):
    """d = torch.cdist(x, e) (p = 2, mm path).  Backward of the same definition:
    G = g / d (0 where d == 0, as torch.cdist's backward defines it);
    dx = x * rowsum(G) - G @ e;  de = e * colsum(G) - G^T @ x   — two plain fp32 GEMMs (library), no [N, K, D] temporary."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        d = ops.distance(x, e, 'L2')
        ctx.save_for_backward(x, e, d)
        return d

    @staticmethod
    def backward(ctx, g):
        x, e, d = ctx.saved_tensors
        x32, e32 = x.float(), e.float()
        G = torch.where(d > 0, g / d, torch.zeros_like(g))
        gx = ge = None
        if ctx.needs_input_grad[0]:
            gx = (x32 * G.sum(1, keepdim=True) - G @ e32).to(x.dtype)
        if ctx.needs_input_grad[1]:
            ge = (e32 * G.sum(0).unsqueeze(1) - G.t() @ x32).to(e.dtype)
        return gx, ge


class _DotMatrix(Function):
    """d = 1 - xn @ en^T on operands that are already normalised (CosineDistance: distances.py:39-46)."""

    @staticmethod
    def forward(ctx, xn: torch.Tensor, en: torch.Tensor, metric: str = 'Cosine') -> torch.Tensor:
        ctx.save_for_backward(xn, en)
        return ops.distance(xn, en, metric)

    @staticmethod
    def backward(ctx, g):
        xn, en = ctx.saved_tensors
        gx = -(g @ en) if ctx.needs_input_grad[0] else None
        ge = -(g.t() @ xn) if ctx.needs_input_grad[1] else None
        return gx, ge, None


class LazyDistance(torch.Tensor):
    """memo['distance'] (vq/algorithms/vq/quantizers.py:98) without the cost: a tensor-typed handle of shape [N, K]
    whose values are produced (by the HIP distance kernel, with autograd to x and the codebook) only when a consumer
    actually touches them.  ``d.argmin(0)`` / ``d.argmin(-1)`` — the only uses by shipped configs — never materialise:
    they run the fused column / row argmin.  Any other torch function or tensor method applied to it (softmax,
    division, indexing, einops.rearrange, torch.cat, ...) transparently sees the real fp32 matrix, so third-party
    callbacks and losses written against the reference keep working."""

    _METADATA = frozenset({'shape', 'dtype', 'device', 'layout', 'ndim', 'requires_grad', 'is_cuda', 'size', 'dim',
                           'numel', 'nelement', 'is_floating_point', 'is_complex', 'element_size', 'stride',
                           'is_contiguous', 'storage_offset', '__len__', 'is_sparse', 'is_quantized', 'is_meta',
                           'names', 'grad_fn', 'grad', 'is_leaf', 'output_nr', '_version', 'data_ptr', 'is_inference',
                           'retains_grad', '_base', 'is_nested', 'is_mkldnn', 'is_xpu', 'is_mps', 'is_cpu'})

    @staticmethod
    def __new__(cls, distance: 'BaseDistance', x: torch.Tensor, e: torch.Tensor, xq: Optional[torch.Tensor] = None,
                eq: Optional[torch.Tensor] = None, metric: Optional[str] = None):
        # under the reference's bf16 autocast the cosine matrix IS a bf16 tensor (distances.py:39-46 on autocast's list)
        dtype = torch.bfloat16 if (metric or distance.metric) == 'CosineBF16' else torch.float32
        return torch.Tensor._make_wrapper_subclass(cls, (x.shape[0], e.shape[0]), dtype=dtype, device=x.device)

    def __init__(self, distance: 'BaseDistance', x: torch.Tensor, e: torch.Tensor,
                 xq: Optional[torch.Tensor] = None, eq: Optional[torch.Tensor] = None, metric: Optional[str] = None) -> None:
        """``xq`` / ``eq``: the latents / the codebook in the form the exact definition consumes (normalised for
        cosine), when the encode that produced this handle has already computed them.  ``metric``: the metric the encode
        ran with — fixed here, so a consumer that touches the handle outside the autocast region of the encode still sees
        that encode's matrix."""
        self._distance, self._x, self._e, self._xq, self._eq = distance, x, e, xq, eq
        self._metric = metric or distance.metric
        self._value: Optional[torch.Tensor] = None

    @property
    def operands(self):
        return self._x, self._e

    @property
    def metric(self) -> str:
        return self._metric

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self._value = self._distance.matrix(self._x, self._e, self._metric)
        return self._value

    def fused_argmin(self, dim: int) -> torch.Tensor:
        if dim == 0:        # NearestAnchor: d.argmin(0) — nearest latent per code
            if self._xq is not None:
                xq = self._xq
                eq = self._eq if self._eq is not None else self._distance.exact_codebook(self._e, self._metric)
            else:
                xq, eq = self._distance.exact_operands(self._x, self._e, self._metric)
            return ops.col_argmin(xq, eq, self._metric)
        if self._metric == self._distance.metric:
            return self._distance.argmin(self._x, self._e)
        # touched outside the autocast region of its encode: the image of that encode's metric
        return self._distance.argmin(self._x, self._e, prepared=self._distance.prepare(self._e, self._metric))

    def __repr__(self):
        return f'LazyDistance({self.metric}, shape={tuple(self.shape)}, materialized={self._value is not None})'

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, '__name__', '')
        if name == '__get__':                                   # attribute descriptors: Tensor.shape.__get__ ...
            name = getattr(getattr(func, '__self__', None), '__name__', '')
        if name in cls._METADATA:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if name == 'argmin' and args and isinstance(args[0], LazyDistance) and args[0]._value is None:
            dim = args[1] if len(args) > 1 else kwargs.get('dim')
            if dim in (0, 1, -1) and not kwargs.get('keepdim', False):
                return args[0].fused_argmin(dim)

        def real(a):
            if isinstance(a, LazyDistance):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(i) for i in a)
            return a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*real(args), **{k: real(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        """Safety net for calls that reach the dispatcher without passing __torch_function__ (C++ callers)."""
        from torch.utils._pytree import tree_map
        unwrap = lambda a: a.materialize() if isinstance(a, LazyDistance) else a   # noqa: E731
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))


def as_distance_tensor(d) -> torch.Tensor:
    return d.materialize() if isinstance(d, LazyDistance) else d


class BaseDistance(nn.Module, ABC):
    metric: str

    @abstractmethod
    def forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        pass

    def matrix(self, x: torch.Tensor, e: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        """``forward`` under a metric fixed by the caller (a LazyDistance materialising after its encode)."""
        return self(x, e)

    def exact_operands(self, x: torch.Tensor, e: torch.Tensor, metric: Optional[str] = None):
        """Operands of the fp32 definition (normalised for cosine)."""
        return x.detach(), e.detach()

    def exact_codebook(self, e: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        """The codebook operand of the fp32 definition alone."""
        return e.detach()

    def metric_for(self, D: int) -> str:
        """The metric of a fused encode over D-dimensional rows (``metric``, unless a distance narrows it by D)."""
        return self.metric

    def prepare(self, e: torch.Tensor, metric: Optional[str] = None) -> ops.PreparedCodebook:
        return ops.prepare_codebook(e, metric or self.metric_for(e.shape[-1]))

    def encode(self, x: torch.Tensor, e: torch.Tensor, hist: Optional[torch.Tensor] = None,
               stash: Optional[dict] = None, zero_hist: bool = False) -> torch.Tensor:
        """``argmin`` for a codebook that has no prepared image yet (it changes every training step).  The shipped
        distances do image, token side and, for cosine, the normalisation of x in one library call (``_fused_encode``);
        a subclass that only customises ``prepare`` / ``argmin`` gets exactly those."""
        if zero_hist and hist is not None:
            hist.zero_()
        return self.argmin(x, e, hist=hist, prepared=self.prepare(e), stash=stash)

    def _fused_encode(self, x: torch.Tensor, e: torch.Tensor, hist: Optional[torch.Tensor],
                      stash: Optional[dict], zero_hist: bool = False) -> torch.Tensor:
        quant, cb, xq = ops.encode(x.detach(), e.detach(), self.metric_for(e.shape[-1]), hist=hist, zero_hist=zero_hist)
        if stash is not None:
            stash['xq'] = xq if xq is not None else x.detach()
            rows = cb.exact_rows()
            stash['eq'] = rows if rows is not None else e.detach()
            stash['prepared'] = cb
            stash['metric'] = ops.metric_name(cb.metric)
        return quant

    def encode_map(self, x_map: torch.Tensor, e: torch.Tensor, hist: Optional[torch.Tensor] = None,
                   stash: Optional[dict] = None, zero_hist: bool = False):
        """``encode`` on the NCHW feature map [B, D, H, W] (ops.encode_map): returns (quant [B*H*W], x_rows [B*H*W, D]) —
        x_rows = the token-major latents the rest of the step works on (L2: a copy in the map's dtype)."""
        quant, cb, xrows, xq = ops.encode_map(x_map.detach(), e.detach(), self.metric_for(e.shape[-1]), hist=hist, zero_hist=zero_hist)
        if stash is not None:
            stash['xq'] = xq if xq is not None else xrows
            stash['eq'] = cb.exact_rows() if xq is not None else e.detach()
            stash['prepared'] = cb
            stash['metric'] = ops.metric_name(cb.metric)
        return quant, xrows

    def argmin(self, x: torch.Tensor, e: torch.Tensor, hist: Optional[torch.Tensor] = None,
               prepared: Optional[ops.PreparedCodebook] = None, stash: Optional[dict] = None) -> torch.Tensor:
        """torch.argmin(self(x, e), -1) without materialising the matrix.  ``stash['xq']`` receives the latents as the
        exact definition consumes them, for consumers of the same batch (NearestAnchor's column argmin)."""
        cb = prepared if prepared is not None else self.prepare(e)
        if stash is not None:
            stash['xq'] = x.detach()
            stash['eq'] = e.detach()
            stash['metric'] = ops.metric_name(cb.metric)
        return ops.argmin(x.detach(), cb, hist=hist)


@VQITQuantizerDistanceRegistry.register_()
class L2Distance(BaseDistance):
    metric = 'L2'

    def forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        """torch.cdist(x, e) (mm path), fp32, differentiable."""
        return _L2Matrix.apply(x, e)

    def encode(self, x, e, hist=None, stash=None, zero_hist=False):
        return self._fused_encode(x, e, hist, stash, zero_hist)


def _bf16_valued(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().float()


def autocast_bf16_active() -> bool:
    """True inside ``torch.autocast('cuda', dtype=torch.bfloat16)`` — the region the reference's AutocastCallback opens
    around every GPU train / validation step (vq/runners/base.py:30-48; bf16 wherever the device supports it, which
    MI355X does).  fp16 autocast has no counterpart in the library: the fp32 definition is used."""
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


@VQITQuantizerDistanceRegistry.register_()
class CosineDistance(BaseDistance):
    """distances.py:35-46, INCLUDING what torch autocast makes of it.  The reference's GPU trainers and validators run
    every step inside ``torch.autocast('cuda', bfloat16)`` (vq/runners/base.py:30-48 appends an AutocastCallback):
    ``F.normalize`` is on autocast's fp32 list, the einsum on its bf16 list — normalisation in fp32, the contraction on
    bf16 operands with a bf16 result, ``1 - s`` in bf16, hence the lowest index among equal bf16 distances (SURVEY.md
    §7-7: 5.7 % of the rows have such ties).  ``autocast`` (extension; configs need not name it):

      'auto' (default) — follow the caller: bf16 semantics (``VQHIP_METRIC_COS_BF16``) while
                         ``torch.is_autocast_enabled('cuda')`` with dtype bfloat16, the fp32 definition otherwise —
                         so an unchanged VQ-KD / CVQ-VAE / cluster config returns the reference's indices in both modes;
      None             — always the fp32 definition;   'bf16' — always the autocast values.

    The products are summed in the fp32 definition's order, so a row can differ from one particular GEMM's summation
    order only where the fp32 sum straddles a bf16 rounding boundary.  Gradients of a materialised matrix are those of
    the fp32 definition (straight through the rounding)."""

    def __init__(self, *args, autocast: Optional[str] = 'auto', **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if autocast not in (None, 'auto', 'bf16'):
            raise ValueError(f"CosineDistance: autocast must be 'auto', None or 'bf16', got {autocast!r}")
        self._autocast = autocast

    def _bf16(self) -> bool:
        if self._autocast == 'auto':
            return autocast_bf16_active()
        return self._autocast == 'bf16'

    @property
    def metric(self) -> str:
        return 'CosineBF16' if self._bf16() else 'Cosine'

    def metric_for(self, D: int) -> str:
        # 'auto' follows the caller only where the library has the bf16-autocast form (D <= 1024, D % 8 == 0: it lives on
        # the proposal image); elsewhere the fp32 definition keeps an unchanged config running as it did outside autocast.
        # An explicit autocast='bf16' is a request and fails loudly in the library instead.
        if self._autocast == 'auto' and not ops.coarse_supported(D):
            return 'Cosine'
        return self.metric

    def _operand(self, t: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        n = ops.normalize_rows(t.detach())
        return _bf16_valued(n) if (metric or self.metric) == 'CosineBF16' else n

    def exact_operands(self, x: torch.Tensor, e: torch.Tensor, metric: Optional[str] = None):
        return self._operand(x, metric), self._operand(e, metric)

    def exact_codebook(self, e: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        return self._operand(e, metric)

    @staticmethod
    def cosine_similarity(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        """normalize(x) @ normalize(e).T, returned as 1 - distance (the distance kernel's own value)."""
        return 1 - _DotMatrix.apply(VF.normalize(x), VF.normalize(e), 'Cosine')

    def matrix(self, x: torch.Tensor, e: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        return self(x, e, metric)

    def forward(self, x: torch.Tensor, e: torch.Tensor, metric: Optional[str] = None) -> torch.Tensor:
        metric = metric or self.metric
        xn, en = VF.normalize(x), VF.normalize(e)
        if metric == 'CosineBF16':              # bf16-valued operands, gradients straight through the rounding
            xn = xn + (_bf16_valued(xn.detach()) - xn.detach())
            en = en + (_bf16_valued(en.detach()) - en.detach())
            return _DotMatrix.apply(xn, en, metric).bfloat16()     # exact: the kernel's values are bf16 values
        return _DotMatrix.apply(xn, en, metric)

    def encode(self, x, e, hist=None, stash=None, zero_hist=False):
        return self._fused_encode(x, e, hist, stash, zero_hist)

    def argmin(self, x, e, hist=None, prepared=None, stash=None):
        cb = prepared if prepared is not None else self.prepare(e)
        xn = self._operand(x, ops.metric_name(cb.metric))
        if stash is not None:
            stash['xq'] = xn
            stash['eq'] = cb.exact_rows()          # normalize(e), as the image made from e holds it
            stash['metric'] = ops.metric_name(cb.metric)
        return ops.argmin(xn, cb, hist=hist)   # the image holds normalize(e)
