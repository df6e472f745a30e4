"""ctypes/numpy front end of ``vq_oracle.c`` (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Every function takes/returns numpy arrays; fp32 in, int64 indices out, exactly as the reference's
tensors (vq/algorithms/vq/quantizers.py:92-108).  The elementwise codebook-update formulas that need no
summation order (EMA, CVQ decay) are restated here in numpy float32.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libvq_oracle.so')
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile vq_oracle.c with the committed Makefile (gcc); returns the .so path."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, 'vq_oracle.c'))
    ):
        subprocess.run(['make', '-C', _HERE, '-B', 'libvq_oracle.so'], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        L = _lib
        L.vqo_row_sqnorm.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.vqo_normalize_rows.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, _f32p]
        for name in ('vqo_l2_dist', 'vqo_cos_dist'):
            getattr(L, name).argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _f32p]
        for name in ('vqo_l2_argmin', 'vqo_cos_argmin', 'vqo_cos_bf16_argmin'):
            getattr(L, name).argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                         _i64p, _f32p]
        for name in ('vqo_col_argmin', 'vqo_row_argmin'):
            getattr(L, name).argtypes = [_f32p, ctypes.c_int64, ctypes.c_int64, _i64p]
        L.vqo_gather_ste.argtypes = [_f32p, _f32p, _i64p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p]
        L.vqo_mse.argtypes = [_f32p, _f32p, ctypes.c_int64]
        L.vqo_mse.restype = ctypes.c_float
        L.vqo_bincount.argtypes = [_i64p, ctypes.c_int64, ctypes.c_int64, _i64p]
        L.vqo_scatter_add_rows.argtypes = [_f32p, _i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _f32p]
        L.vqo_version.restype = ctypes.c_int
    return _lib


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_i64p)


def row_sqnorm(v) -> np.ndarray:
    v = _f(v)
    out = np.empty(v.shape[0], np.float32)
    lib().vqo_row_sqnorm(_fp(v), v.shape[0], v.shape[1], _fp(out))
    return out


def normalize_rows(v, eps: float = 1e-12) -> np.ndarray:
    """F.normalize(v) — vq/algorithms/vq/callbacks/normalize.py:24,27."""
    v = _f(v)
    out = np.empty_like(v)
    lib().vqo_normalize_rows(_fp(v), v.shape[0], v.shape[1], eps, _fp(out))
    return out


def _dist(fn, x, e) -> np.ndarray:
    x, e = _f(x), _f(e)
    assert x.shape[1] == e.shape[1]
    d = np.empty((x.shape[0], e.shape[0]), np.float32)
    getattr(lib(), fn)(_fp(x), _fp(e), x.shape[0], e.shape[0], x.shape[1], _fp(d))
    return d


def l2_dist(x, e) -> np.ndarray:
    """L2Distance.forward — vq/algorithms/vq/distances.py:31-32."""
    return _dist('vqo_l2_dist', x, e)


def cos_dist(x, e) -> np.ndarray:
    """CosineDistance.forward — vq/algorithms/vq/distances.py:38-46."""
    return _dist('vqo_cos_dist', x, e)


def _argmin(fn, x, e, with_min: bool):
    x, e = _f(x), _f(e)
    assert x.shape[1] == e.shape[1]
    idx = np.empty(x.shape[0], np.int64)
    mind = np.empty(x.shape[0], np.float32)
    getattr(lib(), fn)(_fp(x), _fp(e), x.shape[0], e.shape[0], x.shape[1], _ip(idx), _fp(mind))
    return (idx, mind) if with_min else idx


def l2_argmin(x, e, with_min: bool = False):
    """VectorQuantizer._encode with L2Distance — vq/algorithms/vq/quantizers.py:97-99."""
    return _argmin('vqo_l2_argmin', x, e, with_min)


def cos_argmin(x, e, with_min: bool = False):
    """VectorQuantizer._encode with CosineDistance."""
    return _argmin('vqo_cos_argmin', x, e, with_min)


def cos_bf16_argmin(x, e, with_min: bool = False):
    """VectorQuantizer._encode with CosineDistance under the reference's bf16 autocast (vq/runners/base.py:30-48):
    bf16 operands after the fp32 normalisation, bf16 similarity and distance, lowest index on ties."""
    return _argmin('vqo_cos_bf16_argmin', x, e, with_min)


def col_argmin(d) -> np.ndarray:
    """NearestAnchor: d.argmin(0) — vq/algorithms/cvqvae/anchors.py:83."""
    d = _f(d)
    idx = np.empty(d.shape[1], np.int64)
    lib().vqo_col_argmin(_fp(d), d.shape[0], d.shape[1], _ip(idx))
    return idx


def row_argmin(d) -> np.ndarray:
    d = _f(d)
    idx = np.empty(d.shape[0], np.int64)
    lib().vqo_row_argmin(_fp(d), d.shape[0], d.shape[1], _ip(idx))
    return idx


def gather_ste(x, e, idx):
    """_decode + ste — vq/algorithms/vq/quantizers.py:107,116; utils/ste.py:10. Returns (z, x+(z-x))."""
    x, e, idx = _f(x), _f(e), _i(idx)
    z = np.empty_like(x)
    out = np.empty_like(x)
    lib().vqo_gather_ste(_fp(x), _fp(e), _ip(idx), x.shape[0], x.shape[1], _fp(z), _fp(out))
    return z, out


def mse(a, b) -> np.float32:
    """todd MSELoss(mean) as used by CodebookLoss/CommitmentLoss — vq/algorithms/vq/losses.py:50,62."""
    a, b = _f(a), _f(b)
    return np.float32(lib().vqo_mse(_fp(a), _fp(b), a.size))


def vqgan_loss(z, x, beta: float = 0.25) -> np.float32:
    """VQGANLoss.forward: codebook + beta * commitment — vq/algorithms/vq/losses.py:119-127."""
    m = mse(z, x)
    return np.float32(m + np.float32(beta) * m)


def bincount(idx, K: int) -> np.ndarray:
    """QuantStatistics.bin_count — vq/algorithms/vq/utils.py:40-42."""
    idx = _i(idx)
    out = np.empty(K, np.int64)
    lib().vqo_bincount(_ip(idx), idx.shape[0], K, _ip(out))
    return out


def scatter_add_rows(src, idx, K: int) -> np.ndarray:
    """centroids.scatter_add_(0, quant, x) — vq/algorithms/vqkd/quantizers/callbacks.py:60-62."""
    src, idx = _f(src), _i(idx)
    dst = np.zeros((K, src.shape[1]), np.float32)
    lib().vqo_scatter_add_rows(_fp(src), _ip(idx), src.shape[0], K, src.shape[1], _fp(dst))
    return dst


# ---- elementwise codebook-update formulas (no summation order involved) -------------------------

def ema(a, b, decay) -> np.ndarray:
    """todd.utils.ema(a, b, decay) = a*decay + b*(1-decay) (un-vendored; definition fixed in SURVEY §8c)."""
    a, b = _f(a), _f(b)
    decay = np.asarray(decay, np.float32)
    return (a * decay + b * (np.float32(1) - decay)).astype(np.float32)


def kmeans_centroids(x, idx, e, hist=None, sums=None) -> np.ndarray:
    """VQKDCallback._kmeans — vq/algorithms/vqkd/quantizers/callbacks.py:44-71 (hist/sums may be the
    all-reduced ones)."""
    e = _f(e)
    K = e.shape[0]
    if hist is None:
        hist = bincount(idx, K)
    if sums is None:
        sums = scatter_add_rows(x, idx, K)
    occ = hist.reshape(K, 1)
    occurred = occ > 0
    cent = (sums / np.maximum(occ, 1).astype(np.float32)).astype(np.float32)
    return np.where(occurred, cent, e).astype(np.float32)


def cvq_decay(p, K: int, ema_decay: float, eps: float = 1e-3) -> np.ndarray:
    """decay = 1 - exp(-p*K*10/(1-ema.decay) - eps) — vq/algorithms/cvqvae/quantizer_callback.py:98-101."""
    p = _f(p).reshape(-1, 1)
    arg = -p * np.float32(K) * np.float32(10) / np.float32(1 - ema_decay) - np.float32(eps)
    return (np.float32(1) - np.exp(arg, dtype=np.float32)).astype(np.float32)
