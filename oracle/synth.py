"""Deterministic synthetic inputs shared by the fixture generator, the tests and bench.py
(TEST/BENCH INFRASTRUCTURE — numpy PCG64 streams, reproducible on any host; fixtures additionally pin
a sha256 of every regenerated input so a drifting generator is detected, not silently accepted)."""
from __future__ import annotations

import hashlib

import numpy as np


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def normal(seed: int, *shape: int) -> np.ndarray:
    return rng(seed).standard_normal(shape, dtype=np.float32)


def uniform(seed: int, lo: float, hi: float, *shape: int) -> np.ndarray:
    return rng(seed).uniform(lo, hi, shape).astype(np.float32)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 → bf16, returned as bf16-valued fp32 (finite inputs)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def unit_rows(a: np.ndarray) -> np.ndarray:
    n = np.sqrt((a.astype(np.float64) ** 2).sum(-1, keepdims=True))
    return (a / np.maximum(n, 1e-12)).astype(np.float32)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- named input families (seeded) ----------------------------------------------------------------

def make_inputs(kind: str, seed: int, N: int, K: int, D: int):
    """Returns (x[N,D], w[K,D]) fp32 for a named distribution."""
    if kind == 'normal':                 # x, W ~ N(0,1)            
        return normal(seed, N, D), normal(seed + 1, K, D)
    if kind == 'normal_bf16x':           # x bf16-valued (autocast), W fp32 (C2)
        return bf16_round(normal(seed, N, D)), normal(seed + 1, K, D)
    if kind == 'planted':                # x = W[i] + 0.3 eps
        w = normal(seed + 1, K, D)
        pick = rng(seed + 2).integers(0, K, N)
        return (w[pick] + np.float32(0.3) * normal(seed, N, D)).astype(np.float32), w
    if kind == 'vqgan_init':             # W ~ U(-1/K, 1/K): the reference's VQGAN init (stress)
        return normal(seed, N, D), uniform(seed + 1, -1.0 / K, 1.0 / K, K, D)
    if kind == 'unit':                   # unit-norm rows (C3)
        return unit_rows(normal(seed, N, D)), unit_rows(normal(seed + 1, K, D))
    if kind == 'int':                    # small integers: every product/sum exact in fp32
        g = rng(seed)
        w = g.integers(-4, 5, (K, D)).astype(np.float32)
        if K >= 8:
            w[K // 2] = w[1]             # duplicate codes → lowest index must win
            w[K - 1] = w[1]
            w[5] = w[2]
            w[3] = 0                     # a zero code
        pick = g.integers(0, K, N)
        x = (w[pick] + g.integers(-1, 2, (N, D))).astype(np.float32)   # planted, integer noise
        if N >= 4:
            x[0] = w[1]                  # exact hit on a duplicated code (distance 0, clamp path)
            x[1] = 0                     # zero token
            x[2] = w[K - 1] if K >= 8 else x[2]
            x[3] = g.integers(-4, 5, D)  # unplanted row: many near ties in the sqrt domain
        return x, w
    raise ValueError(kind)
