#!/usr/bin/env python3
"""The all-fp32 MFMA route (vqhip_argmin_exact / vqhip_distance / column fallback): time and TFLOP/s at a few shapes.
usage: time_exact_tiled.py   (VQHIP_LIB selects another build)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
g = torch.Generator(device='cuda').manual_seed(7)
for N, K, D, dt in ((65536, 8192, 256, torch.float32), (65536, 8192, 256, torch.bfloat16), (65536, 8192, 100, torch.float32),
                    (16384, 16384, 1280, torch.float32), (100352, 8192, 32, torch.float32), (8192, 16384, 254, torch.float32)):
    x = torch.randn(N, D, device='cuda', generator=g).to(dt)
    w = torch.randn(K, D, device='cuda', generator=g)
    for _ in range(2):
        idx = ops.argmin_exact(x, w, 'L2')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        ops.argmin_exact(x, w, 'L2')
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    print(f'argmin_exact N={N} K={K} D={D} {str(dt)[6:]}: {ms:.3f} ms  {2 * N * K * D / ms / 1e9:.1f} TFLOP/s  idx sha1 {hashlib.sha1(idx.cpu().numpy().tobytes()).hexdigest()[:10]}')
