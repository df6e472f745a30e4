#!/usr/bin/env python3
"""Sweep of the gather kernel's A/B knobs (workgroup cap, streaming mode) in one process: CUDA events around 30 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import _lib, ops
K, D = 16384, 256
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
for N in (65536, 262144, 524288):
    x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
    idx = torch.randint(0, K, (N,), device='cuda', generator=g)
    for nt in (1, 2):
        for cap in (128, 256, 512, 1024):
            L.vqhip_set_tuning(3, cap); L.vqhip_set_tuning(4, nt)
            for _ in range(3): ops.gather_ste_loss(x, w, idx, need_z=False)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): ops.gather_ste_loss(x, w, idx, need_z=False)
            b.record(); torch.cuda.synchronize()
            print(f'N={N:7d} nt={"on " if nt == 1 else "off"} cap={cap:5d}: {a.elapsed_time(b) / 30 * 1e3:8.1f} us', flush=True)
L.vqhip_set_tuning(3, 0); L.vqhip_set_tuning(4, 0)
