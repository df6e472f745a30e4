#!/usr/bin/env python3
"""Per-step difference between the one-call and the hook-by-hook VQ-KD module step (and hook-by-hook against itself)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import synth
import test_gpu_one_call as T

N, K, D = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 1024, 32
w0 = synth.unit_rows(synth.rng(8).standard_normal((K, D), dtype=np.float32))
xs = T.batches(N, K, D, w0, 4, 34)
gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) / (N * D)
runs = []
for one_call in (False, False, True, True):
    q = T.build(T.vqkd_cfg(K, D), w0, no_grad_params=True)
    runs.append(T.run_steps(q, xs, gz, one_call))
def md(a, b): return float((a.float() - b.float()).abs().max())
for name, (a, b) in (('hook vs hook', (0, 1)), ('one vs one', (2, 3)), ('hook vs one', (0, 2))):
    for t in range(4):
        ra, rb = runs[a][t], runs[b][t]
        print(name, 'step', t, 'quant', int((ra['quant'] != rb['quant']).sum()), 'w', md(ra['w'], rb['w']), 'z', md(ra['z'], rb['z']),
              'memo_x', md(ra['memo_x'], rb['memo_x']), 'loss', abs(float(ra['loss']) - float(rb['loss'])), 'gx', md(ra['gx'], rb['gx']))
