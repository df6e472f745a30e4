#!/usr/bin/env python3
"""Timing of vqhip_gather_ste_map against the token-major vqhip_gather_ste_mse (VQHIP_LIB selects an experiment build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
g = torch.Generator(device='cuda').manual_seed(1)
B, D, S, K = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 256, 16, 16384
w = torch.randn(K, D, device='cuda', generator=g)
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for dt in (torch.bfloat16, torch.float32):
    xr = torch.randn(B * S * S, D, device='cuda', generator=g).to(dt)
    idx = torch.randint(0, K, (B * S * S,), device='cuda', generator=g)
    print(f'{str(dt)[6:]:9s} map {timeit(lambda: ops.gather_ste_map(xr, w, idx, B, S, S, 0.25)):7.1f} us   token-major {timeit(lambda: ops.gather_ste_mse(xr, w, idx, beta=0.25)):7.1f} us')
