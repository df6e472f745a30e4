"""The reference path restated with the same ATen ops (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

The reference package cannot be imported in the build container (``import todd`` fails: todd_ai
@ed2a3ae is un-vendored), but its quantizer adds no arithmetic of its own: every line on
the path is a stock PyTorch op.  This module restates those lines one-to-one so that
(a) ``oracle/make_golden.py`` can produce fixtures from the very ops the reference runs,
(b) floating-point results (losses, STE output, codebook updates) have a tolerance reference, and
(c) ``bench.py`` can time "the reference's CPU path" on the GPU box's host cores.
It is never imported by the product package.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# -- distances: vq/algorithms/vq/distances.py --------------------------------------------------------

def l2_distance(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """L2Distance.forward (distances.py:31-32)."""
    return torch.cdist(x, e)


def cosine_similarity(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """CosineDistance.cosine_similarity (distances.py:38-43)."""
    x = F.normalize(x)
    e = F.normalize(e)
    return torch.einsum('x d, e d -> x e', x, e)


def cosine_distance(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """CosineDistance.forward (distances.py:45-46)."""
    return 1 - cosine_similarity(x, e)


DISTANCES = {'L2': l2_distance, 'Cosine': cosine_distance}


# -- quantizer: vq/algorithms/vq/quantizers.py, .../quantizers/base.py, utils/ste.py -----------------

def encode(x: torch.Tensor, w: torch.Tensor, distance: str = 'L2'):
    """VectorQuantizer._encode (quantizers.py:92-100): returns (quant, distance matrix)."""
    d = DISTANCES[distance](x, w.clone())
    return torch.argmin(d, dim=-1), d


def decode(quant: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """VectorQuantizer._decode (quantizers.py:102-108): nn.Embedding gather."""
    return F.embedding(quant, w)


def ste(z: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """utils/ste.py:9-10."""
    return x + (z - x).detach()


def mse_loss(a: torch.Tensor, b: torch.Tensor, norm: bool = False) -> torch.Tensor:
    """todd.models.losses.MSELoss(norm=...) as fixed in SURVEY.md §8c: optional F.normalize(dim=1) of
    both arguments, then mean-reduced squared error with weight 1."""
    if norm:
        a, b = F.normalize(a), F.normalize(b)
    return F.mse_loss(a, b)


def codebook_loss(z, x, norm: bool = False):
    """CodebookLoss.forward (losses.py:44-50)."""
    return mse_loss(z, x.detach(), norm)


def commitment_loss(z, x, norm: bool = False):
    """CommitmentLoss.forward (losses.py:56-62)."""
    return mse_loss(z.detach(), x, norm)


def vqgan_loss(z, x, beta: float = 0.25):
    """VQGANLoss.forward (losses.py:119-127)."""
    return codebook_loss(z, x) + beta * commitment_loss(z, x)


def forward(x: torch.Tensor, w: torch.Tensor, distance: str = 'L2', loss: str = 'vqgan',
            beta: float = 0.25, normalize: bool = False):
    """BaseQuantizer.forward + VectorQuantizer.forward (base.py:173-182, quantizers.py:110-117) for one
    loss entry.  ``normalize`` applies NormalizeCallback.before_encode (callbacks/normalize.py:22-29).
    Returns dict(x, w, quant, z, z_ste, loss)."""
    if normalize:
        x = F.normalize(x)
        w = F.normalize(w)
    quant, _ = encode(x, w, distance)
    z = decode(quant, w)
    if loss == 'vqgan':
        l = vqgan_loss(z, x, beta)
    elif loss == 'commitment_norm':          # configs/vqkd/model.py:23-25
        l = commitment_loss(z, x, norm=True)
    else:
        raise ValueError(loss)
    return dict(x=x, w=w, quant=quant, z=z, z_ste=ste(z, x), loss=l)


# -- statistics and codebook updates -----------------------------------------------------------------

def bin_count(quant: torch.Tensor, K: int) -> torch.Tensor:
    """QuantStatistics.bin_count (vq/algorithms/vq/utils.py:40-42)."""
    return quant.bincount(minlength=K)


def frequency(hist: torch.Tensor, numel) -> torch.Tensor:
    """QuantStatistics.frequency (utils.py:48-52) on (possibly all-reduced) hist / numel."""
    return hist / numel


def ema(a: torch.Tensor, b: torch.Tensor, decay) -> torch.Tensor:
    """todd.utils.ema — definition fixed in SURVEY.md §8c."""
    return a * decay + b * (1 - decay)


def kmeans(x: torch.Tensor, quant: torch.Tensor, e: torch.Tensor, world_hist=None, world_sums=None):
    """VQKDCallback._kmeans (vqkd/quantizers/callbacks.py:44-71); world_* emulate the all-reduces."""
    K, D = e.shape
    occurrences = bin_count(quant, K) if world_hist is None else world_hist
    occurrences = occurrences.reshape(K, 1)
    if world_sums is None:
        centroids = torch.zeros_like(e)
        centroids.scatter_add_(0, quant.reshape(-1, 1).expand(-1, D), x)
    else:
        centroids = world_sums
    occurred = occurrences > 0
    occurrences = occurrences.clamp_min(1)
    centroids = centroids / occurrences
    return centroids.where(occurred, e)


def vqkd_after_encode(x, quant, w, ema_decay: float = 0.99, world_hist=None, world_sums=None):
    """VQKDCallback.after_encode, training branch (callbacks.py:124-128) + _update_embedding (:73-75)."""
    x = F.normalize(x)
    e = kmeans(x, quant, w, world_hist, world_sums)
    e = F.normalize(e)
    e = ema(w, e, ema_decay)
    return F.normalize(e)


def nearest_anchor(x, d):
    """NearestAnchor._anchors (cvqvae/anchors.py:83-84)."""
    indices = d.argmin(0)
    return x[indices], indices


def cvq_after_encode(x, quant, d, w, p, ema_decay: float = 0.99, eps: float = 1e-3,
                     world_hist=None, world_numel=None, world_size: int = 1, other_anchors=None):
    """CVQVAECallback.after_encode, training branch (cvqvae/quantizer_callback.py:85-103) with
    NearestAnchor, sync=False.  ``other_anchors`` (list) emulates the anchors all-reduce/ws
    (anchors.py:65-67).  Returns (new_w, new_p, anchors, indices, decay)."""
    K = w.shape[0]
    hist = bin_count(quant, K) if world_hist is None else world_hist
    numel = quant.new_tensor(quant.numel()) if world_numel is None else world_numel
    freq = frequency(hist, numel)
    p = ema(p, freq, ema_decay)
    anchors, indices = nearest_anchor(x, d)
    if other_anchors is not None:
        anchors = anchors.clone()
        for a in other_anchors:
            anchors = anchors + a
        anchors = anchors / world_size
    decay = 1 - torch.exp(-p.reshape(K, 1) * K * 10 / (1 - ema_decay) - eps)
    new_w = ema(w, anchors, decay)
    return new_w, p, anchors, indices, decay


def vqgan_init(K: int, D: int, generator=None) -> torch.Tensor:
    """VQGANQuantizer._init_weights {'type':'vqgan'} → uniform_(-1/K, 1/K) (vqgan/quantizer.py:14-21)."""
    w = torch.empty(K, D)
    return w.uniform_(-1.0 / K, 1.0 / K, generator=generator)
