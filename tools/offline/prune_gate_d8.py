#!/usr/bin/env python3
"""Gate experiment (build container, numpy): how many (token tile x code tile) pairs could an EXACT pruned search skip at the
LlamaGen tokenizer shape (K = 16384 unit-norm codes in 8-D)?

Codes are grouped into cells of a coarse quantizer (k-means on the codes, C cells), code tiles of 32 codes are cut from the
cell-sorted codebook, tokens are sorted by their nearest cell and cut into wide tiles of 32 tokens.  For a pair (token tile T,
code tile U) with centres cT, cU (means) and radii rT, rU (largest distance of a member to the centre) every score obeys
    x.e = (cT + dx).(cU + de) <= cT.cU + rT |cU| + rU |cT| + rT rU
(rigorous: Cauchy-Schwarz).  The pair can be skipped when that bound is below  L_T - m,  L_T = the smallest best score of the
tile's tokens (known exactly here; a kernel would know a lower bound once it has streamed the tokens' own cell) and m the
row margin (1e-3 at this shape: fp16 residuals).  Prints the prunable fraction for a few cell counts."""
import sys, time
import numpy as np

rng = np.random.default_rng(3407)
K, D, N = 16384, 8, 65536
m = 1e-3
e = rng.standard_normal((K, D)).astype(np.float32); e /= np.linalg.norm(e, axis=1, keepdims=True)
x = rng.standard_normal((N, D)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
best = np.empty(N, np.float32)
for i in range(0, N, 8192):
    best[i:i + 8192] = (x[i:i + 8192] @ e.T).max(1)

def kmeans(v, C, iters=12):
    c = v[rng.choice(len(v), C, replace=False)].copy()
    for _ in range(iters):
        a = np.empty(len(v), np.int64)
        for i in range(0, len(v), 16384):
            a[i:i + 16384] = (v[i:i + 16384] @ c.T).argmax(1)
        for k in range(C):
            sel = v[a == k]
            if len(sel):
                c[k] = sel.mean(0); c[k] /= np.linalg.norm(c[k]) + 1e-12
    return c, a

for C in (256, 1024, 4096):
    t0 = time.time()
    cells, code_cell = kmeans(e, C)
    order = np.argsort(code_cell, kind='stable')
    es = e[order]
    U = es.reshape(K // 32, 32, D)
    cU = U.mean(1); rU = np.linalg.norm(U - cU[:, None], axis=2).max(1); nU = np.linalg.norm(cU, axis=1)
    tok_cell = np.empty(N, np.int64)
    for i in range(0, N, 16384):
        tok_cell[i:i + 16384] = (x[i:i + 16384] @ cells.T).argmax(1)
    torder = np.argsort(tok_cell, kind='stable')
    xs, bs = x[torder], best[torder]
    T = xs.reshape(N // 32, 32, D)
    cT = T.mean(1); rT = np.linalg.norm(T - cT[:, None], axis=2).max(1); nT = np.linalg.norm(cT, axis=1)
    LT = bs.reshape(N // 32, 32).min(1)
    bound = cT @ cU.T + rT[:, None] * nU[None] + rU[None] * nT[:, None] + rT[:, None] * rU[None]
    prunable = (bound < (LT[:, None] - m)).mean()
    print(f'cells {C:5d}: token-tile radius median {np.median(rT):.3f}, code-tile radius median {np.median(rU):.3f}, '
          f'smallest best score per token tile median {np.median(LT):.3f} -> prunable (token tile, code tile) pairs: {prunable * 100:.1f} %'
          f'   [{time.time() - t0:.0f} s]', flush=True)

# Best case for the tiles: spatially coherent tiles from a recursive median split along the direction of largest spread
# (leaves of exactly 32 members), for the codes and for the tokens alike — no coarse quantizer could hand a kernel tighter
# tiles than these.
def bisect_tiles(v):
    idx = [np.arange(len(v))]
    while len(idx[0]) > 32:
        nxt = []
        for ids in idx:
            p = v[ids]
            c = p - p.mean(0)
            w = np.linalg.svd(c, full_matrices=False)[2][0]
            o = np.argsort(c @ w)
            h = len(ids) // 2
            nxt += [ids[o[:h]], ids[o[h:]]]
        idx = nxt
    return np.stack(idx)

t0 = time.time()
U = e[bisect_tiles(e)]; T = x[bisect_tiles(x)]; bt = best[bisect_tiles(x)]
cU = U.mean(1); rU = np.linalg.norm(U - cU[:, None], axis=2).max(1); nU = np.linalg.norm(cU, axis=1)
cT = T.mean(1); rT = np.linalg.norm(T - cT[:, None], axis=2).max(1); nT = np.linalg.norm(cT, axis=1)
LT = bt.min(1)
bound = cT @ cU.T + rT[:, None] * nU[None] + rU[None] * nT[:, None] + rT[:, None] * rU[None]
print(f'median-split tiles: token-tile radius median {np.median(rT):.3f}, code-tile radius median {np.median(rU):.3f} -> prunable pairs: '
      f'{(bound < (LT[:, None] - m)).mean() * 100:.1f} %   [{time.time() - t0:.0f} s]')
