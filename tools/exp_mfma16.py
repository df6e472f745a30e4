#!/usr/bin/env python3
"""Timing-only A/B of the MFMA shape inside the proposal kernel (build/exp/libvqhip_exp.so, -DVQ_EXPERIMENT_MFMA16):
tuning key 1 = 0 → v_mfma_f32_32x32x16_f16 (real kernel), 1 → v_mfma_f32_16x16x32_f16 with the same operand traffic
(garbage results).  Interleaved rounds in one process."""
import ctypes, os, sys
import numpy as np
os.environ['VQHIP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'exp', 'libvqhip_exp.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import _lib, ops
K, D, N = 16384, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
cb = ops.prepare_codebook(w, 'L2')
times = {0: [], 1: []}
for r in range(9):
    for v in (0, 1):
        L.vqhip_set_tuning(1, v)
        L.vqhip_profile_enable(1)
        for _ in range(3):
            ops.argmin(x, cb)
        torch.cuda.synchronize()
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n)); L.vqhip_profile_enable(0)
        if r > 0:
            times[v].append(ms.value / n.value)
for v in (0, 1):
    t = np.array(times[v]); print(('32x32x16' if v == 0 else '16x16x32'), f'median {np.median(t):.4f} ms min {t.min():.4f} -> {2.0*N*K*D/np.median(t)/1e9:.0f} TF')
