#!/usr/bin/env python3
"""Encode time (prepare + argmin) and proposal-kernel time of several libvqhip builds for one shape, alternating
subprocess rounds on one device; checks every build returns the shipped build's indices.
usage: exp_shape.py N K D L2|Cosine lib1.so lib2.so ...   ('shipped' = the in-tree library; 'lib@6=0' adds tuning knobs)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys, time, hashlib
sys.path.insert(0, %r)
import torch
from vector_quantization_amd import _lib, ops
N, K, D, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
L = _lib.lib()
for kv in filter(None, os.environ.get('VQHIP_TUNE', '').split(',')):      # e.g. VQHIP_TUNE=6=0,2=2 (vqhip_set_tuning keys)
    k, v = kv.split('='); L.vqhip_set_tuning(int(k), int(v))
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
one_call = bool(os.environ.get('VQ_EXP_ENCODE'))      # the training-time form: ops.encode on the raw latents (one library call)
if metric == 'Cosine' and not one_call: x = ops.normalize_rows(x)
def enc():
    if one_call:
        return ops.encode(x, w, metric)[0]
    cb = ops.prepare_codebook(w, metric)
    return ops.argmin(x, cb)
for _ in range(8): idx = enc()
torch.cuda.synchronize()
L.vqhip_profile_enable(1)
t0 = time.perf_counter()
for _ in range(30): enc()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 30
ms, n = ctypes.c_double(0), ctypes.c_int64(0)
L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n))
print(t * 1e3, ms.value / n.value, hashlib.sha1(idx.cpu().numpy().tobytes()).hexdigest())
''' % ROOT
shape = sys.argv[1:5]
libs = sys.argv[5:]
res = {l: [] for l in libs}
for r in range(3):
    for l in libs:
        env = dict(os.environ)
        path, _, tune = l.partition('@')                      # "lib.so@6=0,2=2": tuning knobs for that arm
        if path != 'shipped': env['VQHIP_LIB'] = os.path.join(ROOT, path)
        if tune: env['VQHIP_TUNE'] = tune
        out = subprocess.run([sys.executable, '-c', CHILD] + shape, env=env, capture_output=True, text=True)
        try:
            a, b, h = out.stdout.strip().splitlines()[-1].split()
            res[l].append((float(a), float(b), h))
        except Exception:
            print(l, 'FAILED', out.stderr[-400:])
ref = res[libs[0]][0][2] if res[libs[0]] else None
print('shape', shape)
for k, v in res.items():
    if v:
        print(f'{k:40s} encode {np.median([a for a, _, _ in v]):.4f} ms  proposal kernel {np.median([b for _, b, _ in v]):.4f} ms  same_idx={all(h == ref for _, _, h in v)}')
