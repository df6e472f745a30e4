#!/usr/bin/env python3
"""Timing-only A/B of two builds of libvqhip in one process pair is impossible (one .so per process), so this runs the
experiment library (VQHIP_LIB) and the shipped one in alternating subprocess rounds on the same device."""
import ctypes, os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
import torch
from vector_quantization_amd import _lib, ops
K, D, N = 16384, 256, int(sys.argv[1])
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
cb = ops.prepare_codebook(w, 'L2')
for _ in range(5): ops.argmin(x, cb)
torch.cuda.synchronize()
L.vqhip_profile_enable(1)
for _ in range(20): ops.argmin(x, cb)
torch.cuda.synchronize()
ms, n = ctypes.c_double(0), ctypes.c_int64(0)
L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n))
print(ms.value / n.value)
''' % ROOT
N = sys.argv[1] if len(sys.argv) > 1 else '65536'
res = {'shipped': [], 'exp': []}
for r in range(4):
    for name in ('shipped', 'exp'):
        env = dict(os.environ)
        if name == 'exp': env['VQHIP_LIB'] = os.path.join(ROOT, 'build', 'exp', 'libvqhip_exp.so')
        out = subprocess.run([sys.executable, '-c', CHILD, N], env=env, capture_output=True, text=True)
        res[name].append(float(out.stdout.strip().splitlines()[-1]))
for k, v in res.items():
    print(k, 'median %.4f ms' % np.median(v), ['%.4f' % t for t in v])
