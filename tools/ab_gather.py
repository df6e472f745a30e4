#!/usr/bin/env python3
"""A/B of gather_ste_loss between the shipped library and build/exp/libvqhip_exp.so on one device (alternating
subprocess rounds, CUDA events around 30 calls)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from vector_quantization_amd import ops
K, D, N = 16384, 256, int(sys.argv[1])
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
idx = torch.randint(0, K, (N,), device='cuda', generator=g)
for _ in range(5): ops.gather_ste_loss(x, w, idx, need_z=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): ops.gather_ste_loss(x, w, idx, need_z=False)
b.record(); torch.cuda.synchronize()
print(a.elapsed_time(b) / 30)
''' % ROOT
N = sys.argv[1] if len(sys.argv) > 1 else '524288'
res = {'shipped': [], 'exp': []}
for r in range(3):
    for name in ('shipped', 'exp'):
        env = dict(os.environ)
        if name == 'exp': env['VQHIP_LIB'] = os.path.join(ROOT, 'build', 'exp', 'libvqhip_exp.so')
        out = subprocess.run([sys.executable, '-c', CHILD, N], env=env, capture_output=True, text=True)
        res[name].append(float(out.stdout.strip().splitlines()[-1]))
for k, v in res.items():
    print(k, 'median %.4f ms' % np.median(v), ['%.4f' % t for t in v])
