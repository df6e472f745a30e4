#!/usr/bin/env python3
"""Timing-only comparison of several builds of libvqhip (proposal kernel, HIP events) in alternating subprocess rounds on
one device.  usage: exp_abn.py N lib1.so lib2.so ...   ('shipped' = the in-tree library)"""
import os, subprocess, sys
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
for _ in range(8): ops.argmin(x, cb)
torch.cuda.synchronize()
L.vqhip_profile_enable(1)
for _ in range(20): ops.argmin(x, cb)
torch.cuda.synchronize()
ms, n = ctypes.c_double(0), ctypes.c_int64(0)
L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n))
print(ms.value / n.value)
''' % ROOT
N = sys.argv[1]
libs = sys.argv[2:]
res = {l: [] for l in libs}
for r in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != 'shipped': env['VQHIP_LIB'] = os.path.join(ROOT, l)
        out = subprocess.run([sys.executable, '-c', CHILD, N], env=env, capture_output=True, text=True)
        res[l].append(float(out.stdout.strip().splitlines()[-1]))
for k, v in res.items():
    print(f'{k:36s} median {np.median(v):.4f} ms', ['%.4f' % t for t in v])
