#!/usr/bin/env python3
"""Interleaved A/B of one vqhip_set_tuning key (0 vs 1) in one process on the encode step (prepare + argmin); results are
compared bit for bit.  usage: ab_key.py KEY"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import _lib, ops

KEY = int(sys.argv[1])
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)

def one(N, K, D, metric, dtype=torch.float32, rounds=7, reps=20):
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g).to(dtype)
    if metric == 'Cosine':
        x = ops.normalize_rows(x)
    def enc():
        cb = ops.prepare_codebook(w, metric)
        return ops.argmin(x, cb)
    res, times = {}, {0: [], 1: []}
    for r in range(rounds):
        for arm in (0, 1):
            L.vqhip_set_tuning(KEY, arm)
            idx = enc(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): enc()
            torch.cuda.synchronize()
            times[arm].append((time.perf_counter() - t0) / reps)
            res[arm] = idx.clone()
    L.vqhip_set_tuning(KEY, 1)
    m = {a: sorted(times[a])[len(times[a]) // 2] for a in times}
    print(f'N={N:7d} K={K:5d} D={D:4d} {metric:6s}: key{KEY}=0 {m[0]*1e3:7.4f} ms   key{KEY}=1 {m[1]*1e3:7.4f} ms  ({m[0]/m[1]:.3f}x)  same_idx={torch.equal(res[0], res[1])}', flush=True)

one(3072, 16384, 256, 'Cosine')
one(8192, 16384, 256, 'L2', torch.bfloat16)
one(65536, 16384, 256, 'L2', torch.bfloat16)
one(524288, 16384, 256, 'L2', torch.bfloat16, rounds=3, reps=5)
one(100352, 8192, 32, 'Cosine')
one(12544, 8192, 32, 'Cosine')
one(65536, 8192, 32, "Cosine")
one(262144, 8192, 32, "Cosine")
one(524288, 16384, 8, "L2", rounds=3, reps=5)
one(3072, 8192, 32, "Cosine")
one(80000, 8192, 32, "Cosine")
one(90000, 16384, 256, 'L2', torch.bfloat16)
one(40000, 16384, 256, 'L2', torch.bfloat16)
one(150000, 8192, 64, 'L2')
one(1024, 1024, 256, 'L2')
