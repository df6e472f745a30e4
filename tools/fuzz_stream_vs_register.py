#!/usr/bin/env python3
"""Randomised campaign on the GPU: the streamed form of the whole-batch fp32 pass (exact_stream_kernel) against the register form
(exact_tiled_kernel, vqhip_set_tuning key 18 = 0) — row argmin with minima, distance matrix, column argmin — on random shapes,
row dtypes, metrics and data kinds (duplicates, near-ties, tiny scales, non-finite rows and codes).  Any difference is a bug.
usage: fuzz_stream_vs_register.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
g = torch.Generator(device='cuda').manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
L = _lib.lib()
def ri(lo, hi): return int(torch.randint(lo, hi, (1,), generator=g, device='cuda').item())
def both(fn):
    out = []
    for form in (1, 0):
        L.vqhip_set_tuning(18, form); out.append(fn())
    L.vqhip_set_tuning(18, 1)
    return out
def same(a, b):
    if a.dtype.is_floating_point: a, b = a.view(torch.int32), b.view(torch.int32)
    return bool(torch.equal(a, b))
t_end = time.time() + budget
trials = bad = 0
while time.time() < t_end:
    bf = ri(0, 2)
    D = (8 if bf else 4) * ri(1, [9, 33, 65, 161][ri(0, 4)])
    K = [ri(1, 64), ri(64, 4096), 8192, ri(4096, 20000)][ri(0, 4)]
    N = [ri(1, 512), ri(512, 20000), ri(20000, 120000)][ri(0, 3)]
    mode = ri(0, 3)                                  # 0 argmin, 1 distance, 2 column argmin
    if mode == 1:
        K = 4 * max(1, K // 4) if ri(0, 2) else K    # both store paths of the distance epilogue
        if N * K > 1 << 26: N = max(1, (1 << 26) // K)
    if N * K * D > 2e12: N = max(1, int(2e12 / (K * D)))
    metric = 'L2' if ri(0, 3) else 'Cosine'
    kind = ri(0, 7)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g)
    if kind == 1: x = w[torch.randint(0, K, (N,), device='cuda', generator=g)] + 0.02 * x
    elif kind == 2 and K > 1: w[K // 2:] = w[:K - K // 2] * (1 + 1e-4 * ri(0, 3))
    elif kind == 3: w = (torch.rand(K, D, device='cuda', generator=g) * 2 - 1) / K
    elif kind == 4: x *= 10.0 ** ri(-4, 5); w *= 10.0 ** ri(-4, 5)
    elif kind == 5:                                   # non-finite rows and codes
        x[ri(0, N), ri(0, D)] = float('nan'); x[ri(0, N)] = float('inf')
        w[ri(0, K), ri(0, D)] = float('inf'); w[ri(0, K)] = float('nan')
    if bf: x = x.bfloat16()
    if metric == 'Cosine': x, w = ops.normalize_rows(x), ops.normalize_rows(w)
    if mode == 0:
        (i1, m1), (i0, m0) = both(lambda: ops.argmin_exact(x, w, metric, return_min=True))
        ok = same(i1, i0) and same(m1, m0)
    elif mode == 1:
        d1, d0 = both(lambda: ops.distance(x, w, metric)); ok = same(d1, d0)
    else:
        if D % 8 == 0 and D <= 1024: D_note = 'pipeline'      # (supported D take the proposal pipeline in both runs: still a valid trial)
        c1, c0 = both(lambda: ops.col_argmin(x, w, metric)); ok = same(c1, c0)
    trials += 1
    if not ok:
        bad += 1
        print(f'MISMATCH mode {mode} N {N} K {K} D {D} bf16 {bf} {metric} kind {kind}', flush=True)
print(f'{trials} trials, {bad} with mismatches')
