#!/usr/bin/env python3
"""BASELINE.json configs[3]: CVQ-VAE quantizer training step at the per-rank shape (K=16384, D=256, N=12*256 tokens,
cosine, NearestAnchor) — forward + CVQ codebook update + backward through the nn.Module path, single rank (the
all-reduces of the update are no-ops at world size 1).  Also the VQ-KD step (K=8192, D=32, EMA k-means)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import build_quantizer, Config

def run(name, cfg, N, K, D, iters=30):
    g = torch.Generator(device='cuda').manual_seed(3407)
    q = build_quantizer(cfg)
    q.init_weights(Config(type='vqgan'))
    q = q.cuda().train()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.nn.functional.normalize(torch.randn(K, D, device='cuda', generator=g)))
    x = torch.randn(N, D, device='cuda', generator=g).requires_grad_(True)
    def it():
        q.zero_grad(set_to_none=True); x.grad = None
        z, loss, memo = q(x, {})
        (loss + z.mean()).backward()
    for _ in range(5): it()
    ts = []                 # each step on its own (launch to completion) ...
    for _ in range(iters):
        torch.cuda.synchronize(); t1 = time.perf_counter(); it(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
    ts.sort(); lat = ts[len(ts) // 2]
    if os.environ.get('VQ_PER_ITER'):
        print('   per step ms:', ' '.join(f'{v * 1e3:.2f}' for v in ts))
    torch.cuda.synchronize(); t0 = time.perf_counter()          # ... and back to back (the host runs ahead of the GPU)
    for _ in range(iters): it()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    print(f'{name}: N={N} K={K} D={D}: {lat*1e3:.3f} ms per training step, median of {iters} synchronised steps ({N/lat/1e6:.1f} Mtok/s); '
          f'{t*1e3:.3f} ms back to back', flush=True)

emb = lambda K, D: dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
run('CVQ-VAE', dict(type='VQGANQuantizer', embedding=emb(16384, 256), distance=dict(type='CosineDistance'),
                    losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                    callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]), N, 16384, 256)
run('CVQ-VAE sparse anchors', dict(type='VQGANQuantizer', embedding=emb(16384, 256), distance=dict(type='CosineDistance'),
                    losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                    callbacks=[dict(type='CVQVAECallback', ema=dict(), sparse_anchors=True, anchor=dict(type='NearestAnchor'))]), N, 16384, 256)
run('VQ-KD  ', dict(type='VQKDQuantizer', embedding=emb(8192, 32), distance=dict(type='CosineDistance'),
                    losses=dict(vqgan_loss=dict(type='VQGANLoss', mse=dict(norm=True))),
                    callbacks=[dict(type='VQKDCallback', ema=dict())]), 512 * 196 // 8, 8192, 32)
