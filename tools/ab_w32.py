#!/usr/bin/env python3
"""D <= 16 proposal kernel on v_mfma_f32_32x32x16_f16 (knob 11) against the 16x16x32 form, in one process: encode time, path
counters, identical indices.  usage: ab_w32.py [N K D metric]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import _lib, ops
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)
shapes = [(100352, 8192, 32, 'Cosine'), (100352, 8192, 32, 'L2'), (65536, 8192, 32, 'Cosine'), (524288, 16384, 32, 'L2'), (524288, 16384, 8, 'L2'), (524288, 16384, 8, 'Cosine'), (65536, 16384, 8, 'L2'), (65536, 16384, 8, 'Cosine'), (100352, 8192, 16, 'Cosine'), (100352, 8192, 16, 'L2'), (20000, 4096, 16, 'L2')]
def timeit(fn, reps=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for N, K, D, metric in shapes:
    w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
    if D == 8 and metric == 'L2': x, w = ops.normalize_rows(x), ops.normalize_rows(w)
    xq = ops.normalize_rows(x) if metric == 'Cosine' else x
    out = {}
    for key in (0, 1):
        L.vqhip_set_tuning(11, key)
        cb = ops.prepare_codebook(w, metric)
        idx, st = ops.argmin(xq, cb, return_stats=True)
        t = timeit(lambda: ops.argmin(xq, ops.prepare_codebook(w, metric)))
        out[key] = (idx, st.cpu().tolist(), t)
    same = torch.equal(out[0][0], out[1][0])
    print(f'N={N:7d} K={K:5d} D={D:3d} {metric:6s}: 16x16x32 {out[0][2]:.4f} ms paths {out[0][1][:3]}   32x32x16 {out[1][2]:.4f} ms paths {out[1][1][:3]}   ({out[0][2]/out[1][2]:.3f}x) same_idx={same}', flush=True)
L.vqhip_set_tuning(11, 1)
