#!/usr/bin/env python3
"""A/B of the filtered epilogue of the small-D proposal kernel (vqhip_set_tuning key 5), interleaved rounds in one
process; results are compared bit for bit.  Prints ms per encode (prepare + argmin) per shape and arm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import _lib, ops

L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)

def one(N, K, D, metric, dtype=torch.float32, rounds=7, reps=10):
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g).to(dtype)
    if metric == 'Cosine':
        x = ops.normalize_rows(x)
    def enc():
        cb = ops.prepare_codebook(w, metric)
        return ops.argmin(x, cb, return_stats=True)
    res = {}
    times = {0: [], 1: []}
    for r in range(rounds):
        for arm in (0, 1):
            L.vqhip_set_tuning(5, arm)
            idx, st = enc(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): enc()
            torch.cuda.synchronize()
            times[arm].append((time.perf_counter() - t0) / reps)
            res[arm] = (idx.clone(), st.clone())
    L.vqhip_set_tuning(5, 1)
    same = torch.equal(res[0][0], res[1][0])
    m = {a: sorted(times[a])[len(times[a]) // 2] for a in times}
    print(f'N={N:7d} K={K:5d} D={D:4d} {metric:6s}: plain {m[0]*1e3:7.3f} ms  filtered {m[1]*1e3:7.3f} ms  ({m[0]/m[1]:.2f}x)  '
          f'same_idx={same}  stats plain={res[0][1].tolist()} filtered={res[1][1].tolist()}', flush=True)

one(100352, 8192, 32, 'Cosine')
one(100352, 8192, 32, 'L2')
one(524288, 16384, 8, 'L2')
one(65536, 16384, 64, 'L2')
one(65536, 16384, 128, 'L2', torch.bfloat16)
one(12544, 8192, 32, 'Cosine')
one(3072, 16384, 64, 'L2')
