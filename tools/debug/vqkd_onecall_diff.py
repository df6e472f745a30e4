#!/usr/bin/env python3
"""Where do the one-call VQ-KD forward and the chain of separate calls differ?  Intermediate by intermediate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import synth
from vector_quantization_amd import ops, train_step

N, K, D = 3000, 1024, 32
g = synth.rng(8)
w0 = torch.from_numpy(synth.unit_rows(g.standard_normal((K, D), dtype=np.float32))).cuda()
x = torch.from_numpy(g.standard_normal((N, D), dtype=np.float32) * np.float32(0.3)).cuda() + w0[torch.randint(0, K // 8, (N,), device='cuda')]
# chain
wm = ops.normalize_rows(ops.normalize_rows(w0))
xn = ops.normalize_rows(x)
idx, cb, xq = ops.encode(xn, wm, 'Cosine')
x2 = ops.normalize_rows(xn)
hist = ops.hist(idx, K)
sums = ops.scatter_add_rows(x2, idx, K)
e = wm.clone()
ops.vqkd_update_(e, hist.to(torch.int64), sums, 0.99)
# one call
st = train_step.VqkdStepState()
w_mid, w_out = torch.empty_like(w0), torch.empty_like(w0)
out = train_step.vqkd_forward(x, w0, w_mid, w_out, 'Cosine', 0.99, st, exchange=False, world=1, comm=None)
torch.cuda.synchronize()
ws, packed = list(st.arena._bufs.values())[0]
pay = packed[2 * K + 4: 2 * K + 4 + K * D].view(K, D)
cnt = packed[:K] + 65536 * packed[K:2 * K]
def md(a, b): return float((a - b).abs().max())
print('w_mid', md(wm, w_mid), 'xn', md(xn, out['xn']), 'xq', md(xq, out['xq']), 'x2 vs xq', md(x2, xq), 'idx', int((idx != out['idx']).sum()))
print('hist', md(hist.float(), out['hist'].float()), 'counts', md(cnt, hist.float()), 'sums', md(sums, pay), 'w_out', md(e, w_out))
sums2 = ops.scatter_add_rows(x2, idx, K)
e2 = wm.clone(); ops.vqkd_update_(e2, hist.to(torch.int64), sums2, 0.99)
print('chain twice: sums', md(sums, sums2), 'w', md(e, e2))
e3 = wm.clone(); ops.vqkd_update_(e3, hist.to(torch.int64), pay.contiguous(), 0.99)
print('update kernel on the one-call payload vs one-call w_out', md(e3, w_out))
