#!/usr/bin/env python3
"""Which path the rows of a CVQ-VAE training step take (bench.py --workload cvq state after `settle` steps): rows sent to the
second proposal pass / the candidate re-rank / the whole-codebook fp32 pass, for the row argmin and for NearestAnchor's
role-swapped pass over the listed codes.  usage: cvq_paths.py [tokens] [settle]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vector_quantization_amd import ops

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 60
K, D, dev = 16384, 256, torch.device('cuda', 0)
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3407)))
g = torch.Generator(device=dev).manual_seed(3407)
pool = [(w[torch.randint(0, K, (tokens,), device=dev, generator=g)] + 0.05 * torch.randn(tokens, D, device=dev, generator=g))
        for _ in range(max(8, min(128, (1 << 19) // tokens)))]
cb_cfg = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]
q = bench.build_module(bench.quantizer_cfg(K, D, 'Cosine', cb_cfg), dev, w, train=True)
cb = q._callbacks.callbacks[0]
for i in range(settle):
    q(pool[i % len(pool)], {})
    if i in (0, 1, 5, 20, settle - 1):
        x = ops.normalize_rows(pool[(i + 1) % len(pool)])
        wn = q.embedding.weight.detach()
        idx, st = ops.argmin(x, ops.prepare_codebook(wn, 'Cosine'), return_stats=True)
        # duplicates in the codebook: codes whose nearest OTHER code is (nearly) itself
        wu = torch.nn.functional.normalize(wn)
        sim = (wu[:2048] @ wu.T)
        sim[torch.arange(2048), torch.arange(2048)] = -1
        print(f'step {i + 1}: listed codes {cb.last_exchange_rows}; next batch: second pass {int(st[0])}, re-rank {int(st[1])}, '
              f'whole-codebook fp32 {int(st[2])} of {tokens} rows; codes (first 2048) with another code at cosine > 0.9999: '
              f'{int((sim.max(1).values > 0.9999).sum())}')
