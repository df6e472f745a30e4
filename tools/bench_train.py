#!/usr/bin/env python3
"""Quantizer forward+backward (training mode) through the nn.Module path: VQGAN loss, fused decode/STE/loss, HIP backward
(scatter-add codebook gradient).  Prints ms per iteration; run under rocprofv3 for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import build_quantizer, Config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K, D = 16384, 256
g = torch.Generator(device='cuda').manual_seed(3407)
q = build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                         distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
q.init_weights(Config(type='vqgan')); q = q.cuda().train()
with torch.no_grad():
    q.embedding.weight.copy_(torch.randn(K, D, device='cuda', generator=g))
x = torch.randn(N, D, device='cuda', generator=g).bfloat16().requires_grad_(True)
def it():
    q.zero_grad(set_to_none=True); x.grad = None
    z, loss, memo = q(x, {})
    (loss + z.float().mean()).backward()
for _ in range(5): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 20
for _ in range(R): it()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / R
print(f'train fwd+bwd N={N} K={K} D={D}: {t*1e3:.3f} ms  {N/t/1e6:.1f} Mtok/s')
