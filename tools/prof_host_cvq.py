#!/usr/bin/env python3
"""Host-side profile (cProfile) of the eager CVQ-VAE training step: where the Python time between launches goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import build_quantizer, Config

K, D, N = 16384, 256, 3072
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
x = torch.randn(N, D, device=dev, generator=g).requires_grad_(True)
cfg = dict(type='VQGANQuantizer', embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
           distance=dict(type='CosineDistance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')),
           callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])
q = build_quantizer(cfg); q.init_weights(Config(type='vqgan')); q = q.to(dev).train()
with torch.no_grad(): q.embedding.weight.copy_(w)

def step():
    q.embedding.weight.grad = None; x.grad = None
    z, loss, memo = q(x, {})
    (loss + z.mean()).backward()

for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print('eager step ms', (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60); print(s.getvalue()[:9000])
