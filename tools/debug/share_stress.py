#!/usr/bin/env python3
"""A long proposal-kernel launch under GPU sharing (start P copies at once): vqhip_argmin repeated, every result compared with the
register form of the fp32 route (no LDS-DMA in it).  usage: share_stress.py [N] [K] [D] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vector_quantization_amd import ops, _lib
N, K, D, reps = (int(a) for a in (sys.argv[1:5] + ['65536', '16384', '256', '10'][len(sys.argv) - 1:]))
g = torch.Generator(device='cuda').manual_seed(11)
x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
w = torch.randn(K, D, device='cuda', generator=g)
L = _lib.lib()
L.vqhip_set_tuning(18, 0)
ref = ops.argmin_exact(x, w, 'L2')
ref2 = ops.argmin_exact(x, w, 'L2')
L.vqhip_set_tuning(18, 1)
cb = ops.prepare_codebook(w, 'L2')
bad_p = bad_s = 0
for r in range(reps):
    bad_p += int((ops.argmin(x, cb) != ref).sum().item() > 0)
    bad_s += int((ops.argmin_exact(x, w, 'L2') != ref).sum().item() > 0)
print(f'N {N} K {K} D {D}: register form twice equal: {bool(torch.equal(ref, ref2))}; of {reps} repeats, proposal pipeline wrong {bad_p}, streamed fp32 form wrong {bad_s}')
