"""The reference path restated with the same ATen ops (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

The reference's quantizer adds no arithmetic of its own: every line on the path is a stock PyTorch op.  This
module restates those lines one-to-one.  It is PINNED to the reference itself: ``oracle/make_golden.py`` and
``tests/test_reference_pin.py`` execute the reference's own source files (``oracle/ref_import.py``, build
container only) on every fixture case and assert that the functions below return byte-identical tensors.  Because
/root/reference cannot travel, this restatement is what runs on the GPU box as
(a) the floating-point tolerance reference of the GPU tests (losses, gradients, codebook updates), and
(b) ``bench.py``'s timed "reference CPU path" on the GPU box's host cores.
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


# -- whole training steps, in the reference's hook order (pinned byte-for-byte by oracle/make_golden.py) ----------

def _vqkd_before_encode(x, w):
    """NormalizeCallback.before_encode (normalize.py:22-29) with VQKDCallback._update_embedding (callbacks.py:73-75):
    the codebook is normalised twice."""
    return F.normalize(x), F.normalize(F.normalize(w))


def vqkd_train_step(x, w, ema_decay: float = 0.99, world=None):
    """VQKDQuantizer.forward in train mode (configs/vqkd/model.py:20-26): before_encode -> cosine _encode ->
    VQKDCallback.after_encode -> decode from the UPDATED codebook -> CommitmentLoss(norm=True) -> ste.
    ``world`` = list of the other ranks' (x1, quant) emulates the all-reduces of hist and centroids."""
    x = x.clone().requires_grad_(True)
    x1, w0 = _vqkd_before_encode(x, w)
    quant, _ = encode(x1.detach(), w0, 'Cosine')
    x2 = F.normalize(x1.detach())
    K, D = w0.shape
    hist, sums = None, None
    if world is not None:
        hist = bin_count(quant, K)
        sums = torch.zeros_like(w0).scatter_add_(0, quant.reshape(-1, 1).expand(-1, D), x2)
        for ox2, oq in world:
            hist = hist + bin_count(oq, K)
            sums = sums + torch.zeros_like(w0).scatter_add_(0, oq.reshape(-1, 1).expand(-1, D), ox2)
    w_new = vqkd_after_encode(x1.detach(), quant, w0, ema_decay, hist, sums)     # normalises x1 again (callbacks.py:124)
    z = decode(quant, w_new)
    loss = commitment_loss(z, x1, norm=True)
    z_ste = ste(z, x1)
    loss.backward()
    return dict(quant=quant, w_new=w_new, loss=loss.detach(), grad_x=x.grad, z=z_ste.detach(), x2=x2)


def vqkd_train_step_2rank(x, w, ema_decay: float = 0.99):
    """Two ranks on rows r::2, rank 0's view (hist and centroid sums all-reduced: callbacks.py:52,63-64)."""
    parts = []
    for r in range(2):
        x1, w0 = _vqkd_before_encode(x[r::2], w)
        parts.append((F.normalize(x1), encode(x1, w0, 'Cosine')[0]))
    return vqkd_train_step(x[0::2], w, ema_decay, world=[parts[1]])


def cvq_train_steps(x, w, distance: str, ema_decay: float = 0.99, eps: float = 1e-3, steps: int = 2):
    """CVQVAECallback training steps from p = 0 (quantizer_callback.py:60-73 then :75-105), NearestAnchor."""
    p = torch.zeros(w.shape[0])
    outs = []
    for _ in range(steps):
        quant, d = encode(x, w, distance)
        w, p, anchors, indices, decay = cvq_after_encode(x, quant, d, w, p, ema_decay, eps)
        outs.append(dict(quant=quant, col_idx=indices, p=p, w_new=w, anchors=anchors, decay=decay))
    return outs


def cvq_train_step_2rank(x, w, distance: str, ema_decay: float = 0.99, eps: float = 1e-3):
    """Two ranks on rows r::2, first step.  'avg' = NearestAnchor(sync=False): per-rank anchors summed and divided by
    the world size (anchors.py:64-67); 'sync' = NearestAnchor(sync=True): x and d all-gathered in rank order, one global
    column argmin (anchors.py:50-57)."""
    K = w.shape[0]
    halves = [(x[r::2],) + encode(x[r::2], w, distance) for r in range(2)]
    hist = bin_count(halves[0][1], K) + bin_count(halves[1][1], K)
    numel = torch.tensor(halves[0][1].numel() + halves[1][1].numel())
    p0 = torch.zeros(K)
    out = {}
    a_other = nearest_anchor(halves[1][0], halves[1][2])[0]
    w_new, p, _, idx, _ = cvq_after_encode(halves[0][0], halves[0][1], halves[0][2], w, p0, ema_decay, eps,
                                           world_hist=hist, world_numel=numel, world_size=2, other_anchors=[a_other])
    out['avg'] = dict(w_new=w_new, p=p, col_idx=idx)
    xs, ds = torch.cat([halves[0][0], halves[1][0]]), torch.cat([halves[0][2], halves[1][2]])
    w_new, p, _, idx, _ = cvq_after_encode(xs, halves[0][1], ds, w, p0, ema_decay, eps, world_hist=hist, world_numel=numel)
    out['sync'] = dict(w_new=w_new, p=p, col_idx=idx)
    return out


def vqkd_lazy_init(x, w, iters: int = 10):
    """VQKDCallback.lazy_init_weights, single rank, N >= K branch (callbacks.py:92-106,112).  Consumes the global
    ``random`` stream exactly like the reference (seed it first)."""
    import random
    x = F.normalize(x)
    K = w.shape[0]
    indices = random.sample(range(x.shape[0]), K)
    e = x[indices]
    quants = []
    for _ in range(iters):
        w = F.normalize(e)                                     # _update_embedding
        quant, _ = encode(x, w, 'Cosine')
        quants.append(quant)
        e = kmeans(x, quant, w)
    return dict(indices=torch.as_tensor(indices), quants=quants, w=F.normalize(e))


def entropy_loss(x, w, distance: str, temperature: float):
    """EntropyLoss.forward (losses.py:139-153) over the distance matrix of ``distance``; returns loss and gradients."""
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    a = DISTANCES[distance](x, w.clone())
    a = a.reshape(-1, a.shape[-1]) / temperature
    probs = a.softmax(-1)
    log_probs = torch.log_softmax(a + 1e-5, -1)
    avg = probs.mean(0)
    avg_entropy = -torch.sum(avg * torch.log(avg + 1e-5))
    sample_entropy = -torch.mean(torch.sum(probs * log_probs, -1))
    loss = sample_entropy - avg_entropy
    loss.backward()
    return dict(loss=loss.detach(), grad_x=x.grad, grad_w=w.grad)
