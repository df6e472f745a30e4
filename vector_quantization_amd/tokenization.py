"""The callers and data formats either side of the quantizer path:

* the BCHW <-> (BHW)C rearrangement around the quantizer call and the model-level entry points
  ``quantize`` / ``encode_to_quant`` / ``decode_from_quant``
  (vq/tasks/image_tokenization/models/base.py:116-146, vq/tasks/image_reconstruction/models.py:97-108);
* the on-disk token formats of the tokenize runners
  (vq/tasks/image_tokenization/runners/callbacks.py:23-53, tools/tokenize_llamagen.py:65-103);
* the codebook metrics (vq/tasks/image_tokenization/runners/metrics.py:25-73).

Encoders, decoders and connectors are out of scope: these helpers take the latent feature map the reference's
``encode`` produces and hand back what its ``decode`` consumes.
"""
from __future__ import annotations

import pathlib
from typing import Optional, TypedDict

import numpy as np
import torch
import torch.distributed as dist
from torch.autograd import Function

from . import ops
from .utils import get_rank, get_world_size


# ---- 'b c h w -> (b h w) c' and back, as HIP transposes with autograd -----------------------------------------------

class _ToTokens(Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor) -> torch.Tensor:
        b, c, h, w = x.shape
        ctx.shape = (b, c, h, w)
        return ops.transpose_last2(x.reshape(b, c, h * w)).reshape(b * h * w, c)

    @staticmethod
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        return ops.transpose_last2(g.reshape(b, h * w, c)).reshape(b, c, h, w)


class _ToMap(Function):
    @staticmethod
    def forward(ctx, z: torch.Tensor, b: int, h: int, w: int) -> torch.Tensor:
        c = z.shape[-1]
        ctx.shape = (b, c, h, w)
        return ops.transpose_last2(z.reshape(b, h * w, c)).reshape(b, c, h, w)

    @staticmethod
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        return ops.transpose_last2(g.reshape(b, c, h * w)).reshape(b * h * w, c), None, None, None


def is_token_major(x: torch.Tensor) -> bool:
    """True when a [B,C,H,W] tensor is stored channels-last, i.e. its memory already is the [(B H W), C] token matrix."""
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def to_tokens(x: torch.Tensor) -> torch.Tensor:
    """einops 'b c h w -> (b h w) c' (models/base.py:124,140).  A channels-last map (what a 1x1 connector conv
    produces when it runs in ``torch.channels_last``) already IS the token matrix: returned as a zero-copy view;
    an NCHW-contiguous map goes through the HIP transpose."""
    if is_token_major(x):
        b, c, h, w = x.shape
        return x.permute(0, 2, 3, 1).reshape(b * h * w, c)
    return _ToTokens.apply(x)


def to_map(z: torch.Tensor, b: int, h: int, w: int, token_major: bool = False) -> torch.Tensor:
    """einops '(b h w) c -> b c h w' (models/base.py:126).  ``token_major=True`` returns the [B,C,H,W] map as a
    zero-copy channels-last view of ``z`` (logical shape and values identical to the NCHW result; the reference's
    trailing ``.contiguous()`` exists only to give the decoder-side conv dense memory, which channels-last is)."""
    if token_major:
        return z.reshape(b, h, w, z.shape[-1]).permute(0, 3, 1, 2)
    return _ToMap.apply(z, b, h, w)


def _map_route(quantizer, x: torch.Tensor, decode: bool) -> bool:
    """An NCHW-contiguous map can be handed to the quantizer as it is (no transpose kernels either side) when the quantizer
    has the map entry points and nothing in its configuration needs the token matrix earlier (``map_fusable``)."""
    if not hasattr(quantizer, 'map_fusable') or not quantizer.map_fusable(x):
        return False
    return quantizer._fusable() if decode else True


def quantize(quantizer, x: torch.Tensor, memo: dict):
    """BaseModel.quantize (models/base.py:116-128): returns (z [B,C,H,W], q_loss, memo)."""
    from .quantizers.memo import get_memo
    b, _, h, w = x.shape
    quantizer_memo = get_memo(memo, 'quantizer')
    quantizer_memo['x_shape'] = x.shape
    token_major = is_token_major(x)
    if not token_major and _map_route(quantizer, x, decode=True):
        # NCHW map: both rearrangements of models/base.py:124-127 are folded into the quantizer's own kernels
        z, q_loss, memo['quantizer'] = quantizer.forward_map(x, quantizer_memo)
        return z, q_loss, memo
    z, q_loss, memo['quantizer'] = quantizer(to_tokens(x), quantizer_memo)
    return to_map(z, b, h, w, token_major), q_loss, memo


def encode_to_quant(quantizer, x: torch.Tensor, memo: dict):
    """BaseModel.encode_to_quant after the encoder (models/base.py:135-146): returns (quant [B,H,W], memo)."""
    from .quantizers.memo import get_memo
    b, _, h, w = x.shape
    quantizer_memo = get_memo(memo, 'quantizer')
    quantizer_memo['x_shape'] = x.shape
    if not is_token_major(x) and _map_route(quantizer, x, decode=False):
        xt, quant, quantizer_memo = quantizer.encode_map(x, quantizer_memo)
    else:
        xt, quant, quantizer_memo = quantizer.encode(to_tokens(x), quantizer_memo)
    quantizer_memo.update(x=xt, quant=quant)
    memo['quantizer'] = quantizer_memo
    return quant.reshape(b, h, w), memo


def decode_from_quant(quantizer, quant: torch.Tensor, memo: dict, token_major: bool = False):
    """BaseModel.decode_from_quant before the decoder (image_reconstruction/models.py:97-106): quant [B,H,W] →
    z [B,C,H,W] (``token_major=True``: as a zero-copy channels-last view of the gathered rows)."""
    from .quantizers.memo import get_memo
    b, h, w = quant.shape
    from .quantizers.vector_quantizer import VectorQuantizer
    if (not token_major and isinstance(quantizer, VectorQuantizer) and quant.is_cuda and not torch.is_grad_enabled()
            and getattr(quantizer, '_fused', True) and type(quantizer)._decode is VectorQuantizer._decode
            and not quantizer._callbacks.overrides_decode_or_loss()):
        return quantizer.decode_map(quant, get_memo(memo, 'quantizer'))[0], memo     # rows gathered straight into the NCHW map
    z, memo['quantizer'] = quantizer.decode(quant.reshape(-1), get_memo(memo, 'quantizer'))
    return to_map(z, b, h, w, token_major), memo


# ---- on-disk token formats ---------------------------------------------------------------------------------------------

class Tokens(TypedDict):
    """runners/callbacks.py:23-26."""
    id_: list
    category: torch.Tensor
    tokens: torch.Tensor


def save_tokens(work_dir, iter_: int, id_: list, category: torch.Tensor, quant: torch.Tensor, x_shape,
                rank: Optional[int] = None) -> pathlib.Path:
    """TokenizeCallback.after_run_iter (runners/callbacks.py:40-53): ``tokens/{iter}_{rank}.pth`` holding
    Tokens(id_, category, tokens[b, h, w])."""
    token_dir = pathlib.Path(work_dir) / 'tokens'
    token_dir.mkdir(parents=True, exist_ok=True)
    b, _, h, w = x_shape
    tokens = quant.reshape(b, h, w)
    path = token_dir / f'{iter_}_{get_rank() if rank is None else rank}.pth'
    torch.save(Tokens(id_=list(id_), category=category, tokens=tokens), path)
    return path


def load_tokens(path) -> Tokens:
    return torch.load(path, weights_only=False)


def save_llamagen(work_dir, image_size: int, iter_: int, quant: torch.Tensor, category: torch.Tensor,
                  rank: Optional[int] = None, world_size: Optional[int] = None):
    """tools/tokenize_llamagen.py:65-103: one ten-crop image per rank and iteration;
    ``llamagen_tokens/imagenet{S}_codes/{i}.npy`` int64 [1, 10, h*w] and ``…_labels/{i}.npy``,
    i = (iter - 1) * world_size + rank."""
    rank = get_rank() if rank is None else rank
    world_size = get_world_size() if world_size is None else world_size
    token_dir = pathlib.Path(work_dir) / 'llamagen_tokens'
    code_dir = token_dir / f'imagenet{image_size}_codes'
    label_dir = token_dir / f'imagenet{image_size}_labels'
    code_dir.mkdir(parents=True, exist_ok=True)
    label_dir.mkdir(parents=True, exist_ok=True)
    i = (iter_ - 1) * world_size + rank
    codes = quant.reshape((1, 10, -1)).cpu().numpy()
    np.save(code_dir / f'{i}.npy', codes)
    np.save(label_dir / f'{i}.npy', category.cpu().numpy())
    return code_dir / f'{i}.npy', label_dir / f'{i}.npy'


# ---- codebook metrics ----------------------------------------------------------------------------------------------------

class CodebookCounts:
    """CodebookMixin (runners/metrics.py:25-56): accumulates bincount(quant) over a run (HIP histogram), all-reduces
    at summary time."""

    def __init__(self, codebook_size: int) -> None:
        self._codebook_size = codebook_size
        self._counts: Optional[torch.Tensor] = None

    def update(self, quant: torch.Tensor, hist: Optional[torch.Tensor] = None) -> None:
        """``hist`` may be the int32 histogram the fused encode already produced (memo['encode']['hist'])."""
        if hist is None:
            hist = ops.hist(quant.reshape(-1), self._codebook_size)
        hist = hist.to(torch.int64)
        self._counts = hist if self._counts is None else self._counts + hist

    def reduced(self) -> Optional[torch.Tensor]:
        if self._counts is None:
            return None
        counts = self._counts.clone()
        if get_world_size() > 1:
            dist.all_reduce(counts)
        return counts

    def summary(self) -> dict:
        """{'codebook_usage': nonzero/K, 'codebook_ppl': entropy of the code histogram in nats}
        (CodebookUsageMetric._summary, CodebookPPLMetric._summary; docs/pretrained_models.md:47-51 quote these)."""
        counts = self.reduced()
        if counts is None:
            return dict(codebook_usage=0.0, codebook_ppl=0.0)
        m = ops.codebook_metrics(counts).tolist()
        return dict(codebook_usage=float(m[0]), codebook_ppl=float(m[1]))
