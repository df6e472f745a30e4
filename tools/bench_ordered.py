#!/usr/bin/env python3
"""Ordered (deterministic) against atomic codebook-side sums: backward of the quantizer and k-means centroid sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator(device='cuda').manual_seed(1)
for (N, K, D) in [(65536, 16384, 256), (524288, 16384, 256), (3072, 16384, 256), (100352, 8192, 32), (65536, 1024, 256)]:
    w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
    idx = ops.argmin(x, ops.prepare_codebook(w, 'L2'))
    xb = x.bfloat16(); gz = torch.randn(N, D, device='cuda', generator=g); one = torch.ones((), device='cuda')
    for name, o in (('atomic', False), ('ordered', True)):
        tb = timeit(lambda: ops.vq_backward(xb, w, idx, gz, one, one, True, True, ordered=o))
        ts = timeit(lambda: ops.scatter_add_rows(x, idx, K, ordered=o))
        print(f'N={N:7d} K={K:5d} D={D:3d} {name:8s}: backward {tb:8.1f} us   centroid sums {ts:8.1f} us', flush=True)
    to = timeit(lambda: ops.token_order(idx, K))
    print(f'    token_order alone {to:8.1f} us')
