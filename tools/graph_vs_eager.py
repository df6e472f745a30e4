#!/usr/bin/env python3
"""Eager vs captured-graph replay of the encode step (prepare + argmin) for small shapes: how much of the step is host
launch cost / launch gaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops

g = torch.Generator(device='cuda').manual_seed(3407)
def run(N, K, D, metric, dtype=torch.float32):
    w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g).to(dtype)
    if metric == 'Cosine': x = ops.normalize_rows(x)
    def step():
        cb = ops.prepare_codebook(w, metric)
        return ops.argmin(x, cb)
    def step_cached(cb=ops.prepare_codebook(w, metric)):
        return ops.argmin(x, cb)
    def timeit(fn, reps=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    te, tc = timeit(step), timeit(step_cached)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        idx = step()
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2, stream=s):
        idx2 = step_cached()
    tg, tg2 = timeit(gr.replay), timeit(gr2.replay)
    print(f'N={N:6d} K={K:5d} D={D:3d} {metric:6s}: eager {te:.4f} ms (argmin only {tc:.4f})   graph {tg:.4f} ms (argmin only {tg2:.4f})', flush=True)

run(3072, 16384, 256, 'Cosine')
run(8192, 16384, 256, 'L2', torch.bfloat16)
run(12544, 8192, 32, 'Cosine')
run(100352, 8192, 32, 'Cosine')
run(1024, 1024, 256, 'L2')
