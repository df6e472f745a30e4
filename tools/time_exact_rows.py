#!/usr/bin/env python3
"""Duration of the last-resort whole-codebook fp32 pass (exact_kernel, row-list form) as a function of the number of listed
rows: rows with a non-finite entry have no usable bound and take that path.  usage (under rocprofv3 --kernel-trace --stats):
time_exact_rows.py [K] [D]; VQ_ROWS=12 for one list length"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device='cuda').manual_seed(1)
w = torch.randn(K, D, device='cuda', generator=g)
cb = ops.prepare_codebook(w, 'L2')
ROWS = [int(v) for v in os.environ.get('VQ_ROWS', '0,1,8,12,16,32,64,128,256,512,1024').split(',')]
for bad in ROWS:
    x = torch.randn(3072, D, device='cuda', generator=g)
    if bad:
        x[torch.randperm(3072, device='cuda', generator=g)[:bad], 3] = float('inf')
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        idx, st = ops.argmin(x, cb, return_stats=True)
    ev[0].record()
    for _ in range(20):
        ops.argmin(x, cb)
    ev[1].record(); torch.cuda.synchronize()
    print(f'{bad:4d} listed rows (counter {int(st[2])}): argmin {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us')
