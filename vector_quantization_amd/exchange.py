"""The ONE exchange step of a training forward: code-hit histogram, token count and the fp32 per-code rows
(VQ-KD centroid sums, CVQ-VAE anchors) cross the wire in a single SUM all-reduce.

The reference issues one collective per quantity (vq/algorithms/vq/utils.py:34-35 twice, vqkd/quantizers/callbacks.py:63-64,
cvqvae/anchors.py:65-67).  Here the integer counts ride in the float buffer as 16-bit pieces, which a float SUM adds exactly
in any order for up to 256 ranks (include/vqhip.h, "packed fp32 buffer") — so the reduced histogram is bit for bit the int64
all-reduce's, and every rank still applies the identical update to identical inputs (the reference's ``is_sync`` invariant).
Device tensors only: packing and unpacking are libvqhip kernels."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .utils import all_reduce_min, all_reduce_sum, exchanging, get_rank, get_world_size

MAX_WORLD = 256


def packed_bytes(K: int, M: int, D: int) -> int:
    """Bytes one rank contributes to the exchange of M rows of D floats next to a K-code histogram."""
    return 4 * ops.pack_floats(K, M, D)


def all_reduce_packed(hist: torch.Tensor, numel: int, sums: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """SUM over the ranks of (hist int32|int64 [K], numel, sums fp32 [K, D]) in ONE collective.
    Returns (hist int64 [K], numel as a device int64 scalar, sums [K, D]) — no host synchronisation."""
    world = get_world_size()
    if world > MAX_WORLD:          # callers route such groups through utils.all_reduce_statistics (int64 counts, no rank limit)
        raise RuntimeError(f'packed exchange: exact fp32 count sums hold for up to {MAX_WORLD} ranks, got {world}')
    K, D = sums.shape
    packed = torch.empty(ops.pack_floats(K, K, D), dtype=torch.float32, device=sums.device)
    head = 2 * K + 4
    packed[head:].view(K, D).copy_(sums)
    ops.pack_counts(hist.contiguous(), int(numel), packed)
    if exchanging():
        all_reduce_sum(packed)
    counts = ops.unpack_counts(packed, K)
    return counts[:K], counts[K], packed[head:].view(K, D)


def cvq_exchange(hist32: torch.Tensor, numel: int, x: torch.Tensor, col_idx: Optional[torch.Tensor], count: Optional[torch.Tensor],
                 cap: int, K: int) -> torch.Tensor:
    """The CVQ-VAE exchange: this rank's histogram, token count and the anchors of the ``cap`` listed codes, all-reduced in
    one collective of 4·(2K + 4 + cap·D) bytes.  Returns the reduced packed buffer (consumed by ``ops.cvq_apply``)."""
    if get_world_size() > MAX_WORLD:
        raise RuntimeError(f'packed exchange: exact fp32 count sums hold for up to {MAX_WORLD} ranks')
    packed = ops.cvq_pack(hist32, numel, x, col_idx, count, cap, K)
    all_reduce_sum(packed)
    return packed


def cvq_exchange_sync(hist32: torch.Tensor, numel: int, x: torch.Tensor, xq: Optional[torch.Tensor], eq: Optional[torch.Tensor],
                      rows: torch.Tensor, col_idx: Optional[torch.Tensor], count: Optional[torch.Tensor], cap: int, K: int,
                      metric) -> torch.Tensor:
    """The CVQ-VAE exchange under NearestAnchor(sync=True) (vq/algorithms/cvqvae/anchors.py:50-57,83-84): every rank's column
    pass covered its own tokens (``col_idx``); the ranks agree on the global nearest latent per listed code through a MIN
    all-reduce of 8·cap bytes of keys, the winner alone contributes its latent to the packed buffer (-0.0 elsewhere) and one SUM
    all-reduce of 4·(2K + 4 + cap·D) bytes delivers histogram, token count and the winners' rows — against the reference's
    all-gather of world·N·(K + D) floats.  ``xq`` / ``eq``: the operands the column pass was given.  Returns the reduced buffer
    (``ops.cvq_apply`` with world = 1: the anchors are not averaged)."""
    if get_world_size() > MAX_WORLD:
        raise RuntimeError(f'packed exchange: exact fp32 count sums hold for up to {MAX_WORLD} ranks')
    rank = get_rank()
    keys = None
    if cap > 0:
        keys = ops.cvq_col_keys(xq, eq, rows, count, cap, col_idx, metric, rank)
        all_reduce_min(keys)
    packed = ops.cvq_pack_sync(hist32, numel, x, keys, count, cap, K, rank)
    all_reduce_sum(packed)
    return packed
