#!/usr/bin/env python3
"""One shape of the encode path, repeated: the program to put after `rocprofv3 --kernel-trace --stats --`.
usage: prof_shape.py N K D L2|Cosine [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops

N, K, D, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
x = torch.randn(N, D, device='cuda', generator=g)
if metric == 'Cosine':
    x = ops.normalize_rows(x)
for _ in range(reps):
    if os.environ.get('VQ_PROF_EXACT'):            # the all-fp32 route instead
        idx = ops.argmin_exact(x, w, metric)
        continue
    cb = ops.prepare_codebook(w, metric)
    idx = ops.argmin(x, cb)
torch.cuda.synchronize()
print('done', int(idx[0]))
