#!/usr/bin/env python3
"""A/B the proposal-kernel slice counts in ONE process with interleaved rounds (cdna guide §5.4 rule 24).
Prints median / min kernel milliseconds per variant (HIP events around the coarse kernel) and TFLOP/s."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import _lib, ops  # noqa: E402

K, D = 16384, 256
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
cb = ops.prepare_codebook(w, 'L2')
variants = [(1, ns) for ns in (0, 1, 2, 4, 8)]
ref = None
times = {v: [] for v in variants}
for r in range(ROUNDS + 1):
    for v in variants:
        L.vqhip_set_tuning(2, v[1])
        L.vqhip_profile_enable(1)
        for _ in range(3):
            idx = ops.argmin(x, cb)
        torch.cuda.synchronize()
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n))
        L.vqhip_profile_enable(0)
        if ref is None:
            ref = idx.clone()
        assert torch.equal(idx, ref), f'variant {v} changed the result'
        if r > 0:
            times[v].append(ms.value / n.value)
flops = 2.0 * N * K * D
for v in variants:
    t = np.array(times[v])
    print(f'nslices={v[1]} (0 = automatic): median {np.median(t):.4f} ms  min {t.min():.4f} ms  '
          f'-> {flops / np.median(t) / 1e9:.0f} TFLOP/s (median)')
