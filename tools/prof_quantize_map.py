#!/usr/bin/env python3
"""quantize() on an NCHW-contiguous / channels-last latent map: timing, or a kernel trace under rocprofv3.
usage: prof_quantize_map.py [B] [D] [HxW side] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import build_quantizer, Config, tokenization as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
g = torch.Generator(device='cuda').manual_seed(3407)
q = build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=C),
                         distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
q.init_weights(Config(type='vqgan')); q = q.cuda().eval()
def timeit(fn, reps=30, warm=5):
    for _ in range(max(warm, reps)): fn()     # same queue depth once untimed: see tools/bench_shapes.py
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
with torch.no_grad():
    q.embedding.weight.copy_(torch.randn(K, C, device='cuda', generator=g))
    for dt in (torch.bfloat16, torch.float32):
        xm = torch.randn(B, C, S, S, device='cuda', generator=g).to(dt)
        for name, fmt in (('NCHW', torch.contiguous_format), ('channels-last', torch.channels_last)):
            xi = xm.contiguous(memory_format=fmt)
            t = timeit(lambda: T.quantize(q, xi, {}))
            print(f'quantize() {name:14s} {str(dt)[6:]:9s} B={B} {C}x{S}x{S} K={K}: {t*1e3:.3f} ms  {B*S*S/t/1e6:.1f} Mtok/s', flush=True)
