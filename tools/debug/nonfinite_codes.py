#!/usr/bin/env python3
"""A code row holding +inf: what the fp32 route returns for rows whose distance to it is NaN (inf - inf)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co, synth
from vector_quantization_amd import ops
x, w = synth.make_inputs('normal', 5, 700, 900, 64)
w[7, 1] = np.inf; w[400, :] = np.nan
xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
idx, dmin = ops.argmin_exact(xd, wd, 'L2', return_min=True)
d = ops.distance(xd, wd, 'L2').cpu().numpy()
ref, refmin = co.l2_argmin(x, w, with_min=True)
dref = co.l2_dist(x, w)
en, xn = ops.row_sqnorm(wd).cpu().numpy(), ops.row_sqnorm(xd).cpu().numpy()
print('en[7], en[400]', en[7], en[400])
idx = idx.cpu().numpy()
print('mismatching rows', np.nonzero(idx != ref)[0][:20], 'of', (idx != ref).sum())
for r in list(range(28, 40)) + [127, 128, 129, 300]:
    print(r, 'x1>0' if x[r, 1] > 0 else 'x1<0', 'gpu', idx[r], float(dmin[r]), 'd[7] gpu/oracle', d[r, 7], dref[r, 7], 'd[400]', d[r, 400], dref[r, 400], 'ref', ref[r])
cb = ops.prepare_codebook(wd, 'L2')
pidx = ops.argmin(xd, cb).cpu().numpy()
print('pipeline (vqhip_argmin) mismatches vs oracle:', (pidx != ref).sum(), ' fp32 route:', (idx != ref).sum())
e1, _, _ = ops.encode(xd, wd, 'L2')
print('one-call encode mismatches vs oracle:', (e1.cpu().numpy() != ref).sum())
