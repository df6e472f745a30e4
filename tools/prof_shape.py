#!/usr/bin/env python3
"""One shape of the encode path, repeated: the program to put after `rocprofv3 --kernel-trace --stats --`.
usage: prof_shape.py N K D L2|Cosine [reps]      VQ_PROF_ENCODE=1: the one-call training-time form (ops.encode: the
unnormalised latents in, image + token side + argmin in one library call); VQ_PROF_BF16=1: bf16 latents"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops

N, K, D, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
x = torch.randn(N, D, device='cuda', generator=g)
if os.environ.get('VQ_PROF_BF16'):
    x = x.bfloat16()
one_call = bool(os.environ.get('VQ_PROF_ENCODE'))
if metric == 'Cosine' and not one_call:
    x = ops.normalize_rows(x)
for _ in range(reps):
    if one_call:
        idx = ops.encode(x, w, metric)[0]
        continue
    if os.environ.get('VQ_PROF_EXACT'):            # the all-fp32 route instead
        idx = ops.argmin_exact(x, w, metric)
        continue
    cb = ops.prepare_codebook(w, metric)
    idx = ops.argmin(x, cb)
torch.cuda.synchronize()
print('done', int(idx[0]))
