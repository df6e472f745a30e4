"""Autograd-aware building blocks of the quantizer path.  Forward and backward arithmetic runs in libvqhip;
torch.autograd.Function is only the glue that hooks the HIP kernels into PyTorch's graph."""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import ops


def _as2d(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


class _Embedding(Function):
    """z = W[idx] (nn.Embedding, vq/algorithms/vq/quantizers.py:107); backward = dense scatter-add into W."""

    @staticmethod
    def forward(ctx, weight: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        flat = idx.reshape(-1)
        ctx.save_for_backward(flat)
        ctx.shape = weight.shape
        z = ops.gather_rows(weight, flat)
        return z.view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        K, D = ctx.shape
        gw = ops.scatter_add_rows(_as2d(g), flat, K) if ctx.needs_input_grad[0] else None
        return gw, None


def embedding(weight: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return _Embedding.apply(weight, idx)


class _MSE(Function):
    """mean((a-b)^2) (todd MSELoss, mean reduction; vq/algorithms/vq/losses.py:50,62)."""

    @staticmethod
    def forward(ctx, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(a, b)
        sse = ops.sse(a, b)
        return (sse / a.numel()).float().reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = ops.diff_scale(a, b, 2.0 / a.numel(), g).to(a.dtype)
        if ctx.needs_input_grad[1]:
            gb = ops.diff_scale(b, a, 2.0 / a.numel(), g).to(b.dtype)
        return ga, gb


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _MSE.apply(a, b)


class _Normalize(Function):
    """F.normalize(v, dim=1, eps=1e-12) in the oracle's summation order."""

    @staticmethod
    def forward(ctx, v: torch.Tensor, eps: float) -> torch.Tensor:
        ctx.save_for_backward(v)
        ctx.eps = eps
        return ops.normalize_rows(_as2d(v), eps).view(v.shape)

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        gv = ops.normalize_rows_bwd(_as2d(v), _as2d(g), ctx.eps).view(v.shape).to(v.dtype)
        return gv, None


def normalize(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    return _Normalize.apply(v, eps)


class _STE(Function):
    """x + (z - x).detach() (vq/tasks/image_tokenization/models/quantizers/utils/ste.py:9-10)."""

    @staticmethod
    def forward(ctx, z: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        ctx.xdtype = x.dtype
        return ops.ste(x, z)

    @staticmethod
    def backward(ctx, g):
        return None, g.to(ctx.xdtype)


def ste(z: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return _STE.apply(z, x)


class _FusedDecodeLoss(Function):
    """decode + straight-through + both MSE terms in one pass:
        z = W[idx];  z_ste = x + sg(z - x);  m_cb = mse(z, sg x);  m_cm = mse(sg z, x)   (same value, two graph nodes)
        combined = m_cb + beta * m_cm   (VQGANLoss, losses.py:119-127 — finished inside the kernel)
    Backward is one fused kernel (vqhip_vq_backward_ex): m_cb's gradient flows to W, m_cm's and z_ste's to x, the combined
    value's to both; gradients of unused outputs are not materialised."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, idx: torch.Tensor, beta: float = 0.0):
        ctx.set_materialize_grads(False)
        ctx.beta = float(beta)
        if x.numel() == 0:
            _, z_ste, sse = ops.gather_ste_loss(x, weight, idx, need_z=False, need_ste=True, need_sse=True)
            ctx.save_for_backward(x, weight, idx)
            m = (sse / x.numel()).float().reshape(())                  # nan, as mse_loss of an empty tensor
            return z_ste.view(x.shape), m, m.clone(), m + beta * m
        _, z_ste, mse = ops.gather_ste_mse(x, weight, idx, beta=beta)   # means and their combination finished inside the kernel
        ctx.save_for_backward(x, weight, idx)
        return z_ste.view(x.shape), mse[0], mse[1], mse[2]

    @staticmethod
    def backward(ctx, g_zste, g_cb, g_cm, g_comb):
        x, weight, idx = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx, gw = ops.vq_backward(x, weight, idx, g_zste, g_cb, g_cm, need_x, need_w, g_comb=g_comb, beta=ctx.beta)
        if gx is not None:
            gx = gx.view(x.shape).to(x.dtype)
        return gx, gw, None, None


def fused_decode_loss(x: torch.Tensor, weight: torch.Tensor, idx: torch.Tensor, beta: float = 0.0):
    """Returns (z_ste, m_cb, m_cm, m_cb + beta*m_cm): the straight-through output, the codebook / commitment MSE values and
    their VQGAN combination."""
    return _FusedDecodeLoss.apply(x, weight, idx, beta)


class _FusedMapDecodeLoss(Function):
    """``_FusedDecodeLoss`` for a quantizer call on the NCHW feature map (tokenization.quantize): the straight-through
    output is written directly as the map [B, D, H, W], the loss values are those of the token form.  ``x_rows`` are the
    token-major rows the map encode produced (``ops.encode_map``); gradients flow to ``x_map`` and the codebook."""

    @staticmethod
    def forward(ctx, x_map: torch.Tensor, x_rows: torch.Tensor, weight: torch.Tensor, idx: torch.Tensor, beta: float = 0.0):
        ctx.set_materialize_grads(False)
        ctx.beta = float(beta)
        b, d, h, w = x_map.shape
        ctx.map_shape, ctx.map_dtype = (b, d, h, w), x_map.dtype
        z_map, mse = ops.gather_ste_map(x_rows, weight, idx, b, h, w, beta=beta)
        ctx.save_for_backward(x_rows, weight, idx)
        return z_map, mse[0], mse[1], mse[2]

    @staticmethod
    def backward(ctx, g_map, g_cb, g_cm, g_comb):
        x_rows, weight, idx = ctx.saved_tensors
        b, d, h, w = ctx.map_shape
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        if ops.backward_map_supported(d, h * w) and ctx.map_dtype in (torch.float32, torch.bfloat16):
            # both rearrangements folded into the kernel: the upstream gradient is read as the map, the latents' gradient written
            # as the map in the map's dtype (vqhip_vq_backward_map); the codebook gradient from the token-major rows as before
            gx = ops.vq_backward_map(x_rows, weight, idx, (b, d, h, w), g_map, g_cm, g_comb, ctx.beta, ctx.map_dtype) if need_x else None
            gw = ops.vq_backward(x_rows, weight, idx, None, g_cb, g_cm, False, True, g_comb=g_comb, beta=ctx.beta)[1] if need_w else None
            return gx, None, gw, None, None
        g_tok = None
        if g_map is not None:                                    # 'b c h w -> (b h w) c' of the incoming gradient
            g_tok = ops.transpose_last2(g_map.float().contiguous().reshape(b, d, h * w)).reshape(b * h * w, d)
        gx, gw = ops.vq_backward(x_rows, weight, idx, g_tok, g_cb, g_cm, need_x, need_w, g_comb=g_comb, beta=ctx.beta)
        if gx is not None:
            gx = ops.transpose_last2(gx.reshape(b, h * w, d)).reshape(b, d, h, w).to(ctx.map_dtype)
        return gx, None, gw, None, None


def fused_map_decode_loss(x_map: torch.Tensor, x_rows: torch.Tensor, weight: torch.Tensor, idx: torch.Tensor, beta: float = 0.0):
    """Returns (z_map [B, D, H, W], m_cb, m_cm, m_cb + beta*m_cm)."""
    return _FusedMapDecodeLoss.apply(x_map, x_rows, weight, idx, beta)


class _Computed:
    """Tensors a one-call training forward (train_step.py) has already produced, handed to the autograd wrappers below
    without becoming graph inputs."""

    def __init__(self, **tensors) -> None:
        self.__dict__.update(tensors)


class _PrecomputedDecodeLoss(Function):
    """``_FusedDecodeLoss`` whose forward values were computed by vqhip_cvq_forward in the same library call as the encode and
    the codebook update: the node only ties them into the graph.  Backward is the same fused kernel."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, done: _Computed, beta: float):
        ctx.set_materialize_grads(False)
        ctx.beta = float(beta)
        ctx.save_for_backward(x, weight, done.idx)
        mse = done.mse
        return done.z_ste.view(x.shape), mse[0], mse[1], mse[2]

    @staticmethod
    def backward(ctx, g_zste, g_cb, g_cm, g_comb):
        x, weight, idx = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx, gw = ops.vq_backward(x, weight, idx, g_zste, g_cb, g_cm, need_x, need_w, g_comb=g_comb, beta=ctx.beta)
        if gx is not None:
            gx = gx.view(x.shape).to(x.dtype)
        return gx, gw, None, None


def precomputed_decode_loss(x: torch.Tensor, weight: torch.Tensor, done: _Computed, beta: float = 0.0):
    """(z_ste, m_cb, m_cm, m_cb + beta*m_cm) of ``fused_decode_loss`` from values a one-call forward already holds."""
    return _PrecomputedDecodeLoss.apply(x, weight, done, beta)


class _VqkdStep(Function):
    """Autograd node of the VQ-KD one-call forward: outputs xn = F.normalize(x) (what memo['x'] holds), the straight-through
    output xn + sg(z - xn) and the commitment loss mean((F.normalize(sg z) - F.normalize(xn))^2) (CommitmentLoss with
    mse norm=True); backward is ONE kernel (vqhip_vqkd_backward).  The codebook receives no gradient: the commitment term
    detaches z and the straight-through output detaches (z - xn)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, done: _Computed):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, done.xn, weight, done.idx)
        xn = done.xn.view(x.shape)
        if not ctx.needs_input_grad[0]:
            ctx.mark_non_differentiable(xn)
        return xn, done.z_ste.view(x.shape), done.mse[0]

    @staticmethod
    def backward(ctx, g_xn, g_zste, g_loss):
        from . import train_step
        x, xn, weight, idx = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        g = g_zste
        if g_xn is not None:                                    # a consumer of memo['x'] other than the straight-through output
            g = g_xn if g is None else g + g_xn
        gx = train_step.vqkd_backward(_as2d(x), xn, weight, idx, None if g is None else _as2d(g), g_loss)
        return gx.view(x.shape).to(x.dtype), None, None


def vqkd_step(x: torch.Tensor, weight: torch.Tensor, done: _Computed):
    """(xn, z_ste, commitment loss) tied into the autograd graph."""
    return _VqkdStep.apply(x, weight, done)


class _VqStep(Function):
    """Autograd node of the one-call forward of a quantizer without an update callback, or with NormalizeCallback alone
    (vqhip_vq_forward): outputs (xn = F.normalize(x) when normalised, else x itself is what memo['x'] holds), the straight-through
    output, both MSE values and their VQGAN combination.  Backward: the fused kernel of ``_FusedDecodeLoss`` on the rows the
    forward quantized, then F.normalize's backward when the rows were normalised."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, done: _Computed, beta: float):
        ctx.set_materialize_grads(False)
        ctx.beta = float(beta)
        ctx.normalized = done.xn is not None
        mse = done.mse
        if ctx.normalized:
            ctx.save_for_backward(x, done.xn, weight, done.idx)
            xn = done.xn.view(x.shape)
            if not ctx.needs_input_grad[0]:                      # F.normalize(x) of latents without a graph has none either
                ctx.mark_non_differentiable(xn)
            return xn, done.z_ste.view(x.shape), mse[0], mse[1], mse[2]
        ctx.save_for_backward(x, weight, done.idx)
        return None, done.z_ste.view(x.shape), mse[0], mse[1], mse[2]

    @staticmethod
    def backward(ctx, g_xn, g_zste, g_cb, g_cm, g_comb):
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if ctx.normalized:
            x, xn, weight, idx = ctx.saved_tensors
            rows = xn
        else:
            x, weight, idx = ctx.saved_tensors
            rows = _as2d(x)
        gr, gw = ops.vq_backward(rows, weight, idx, None if g_zste is None else _as2d(g_zste), g_cb, g_cm, need_x, need_w,
                                 g_comb=g_comb, beta=ctx.beta)
        gx = None
        if need_x:
            if ctx.normalized:
                if g_xn is not None:                             # a consumer of memo['x'] other than the decode / loss tail
                    gr = _as2d(g_xn).float() if gr is None else gr + _as2d(g_xn)
                gx = ops.normalize_rows_bwd(_as2d(x), gr, 1e-12).view(x.shape).to(x.dtype)
            else:
                gx = gr.view(x.shape).to(x.dtype)
        return gx, gw, None, None


def vq_step(x: torch.Tensor, weight: torch.Tensor, done: _Computed, beta: float = 0.0):
    """(xn or None, z_ste, m_cb, m_cm, m_cb + beta*m_cm) of a one-call forward, tied into the autograd graph."""
    return _VqStep.apply(x, weight, done, beta)
