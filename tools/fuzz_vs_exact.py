#!/usr/bin/env python3
"""Randomised campaign on the GPU: vqhip_argmin (fp16 proposals + exact re-rank) against vqhip_argmin_exact (all-fp32
MFMA, itself bit-equal to the CPU oracle in tests/) at sizes the CPU oracle cannot reach.  Any mismatch is a bug.
usage: fuzz_vs_exact.py [seconds] [seed]   (VQ_FUZZ_DIMS=8,16: only those D; VQ_FUZZ_SMALL_N=1: 64 .. 16 384 rows against >= 4096 codes;
VQ_FUZZ_FORCE_EXACT=1: see below; VQ_FUZZ_BF16=1: bf16 latents in every trial — with VQ_FUZZ_DIMS=256 the batches of more than
16 384 rows take the proposal kernel that makes its own token fragments)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
DIMS = [int(d) for d in os.environ['VQ_FUZZ_DIMS'].split(',')] if 'VQ_FUZZ_DIMS' in os.environ else \
    [8, 16, 32, 64, 128, 256, 256, 256, 512, 768, 1024]
if 'VQ_FILTER' in os.environ:                      # A/B knob: 0 = unfiltered epilogue on the small-D proposal kernels
    from vector_quantization_amd import _lib
    _lib.lib().vqhip_set_tuning(5, int(os.environ['VQ_FILTER']))
g = torch.Generator(device='cuda').manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 20261003)
def ri(lo, hi): return int(torch.randint(lo, hi, (1,), generator=g, device='cuda').item())
t_end = time.time() + budget
trials = bad = 0
while time.time() < t_end:
    D = DIMS[ri(0, len(DIMS))]
    K = [ri(1, 64), ri(64, 4096), 8192, 16384, ri(4096, 20000)][ri(0, 5)]
    N = [ri(1, 512), ri(512, 70000), 65536, ri(70000, 300000)][ri(0, 4)]
    if os.environ.get('VQ_FUZZ_SMALL_N') == '1':        # batches that decide inside the proposal kernel over several codebook slices
        N = ri(64, 16385)
        K = [8192, 16384, ri(4096, 20000)][ri(0, 3)]
    if N * K * D > 3e12: N = max(1, int(3e12 / (K * D)))
    metric = 'L2' if ri(0, 3) else ('Cosine' if ri(0, 3) else 'CosineBF16')     # CosineBF16: the bf16-autocast semantics (opt-in)
    kind = ri(0, 7)
    scale = 10.0 ** ri(-4, 5)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g)
    if kind == 1:   # latents near codes
        x = w[torch.randint(0, K, (N,), device='cuda', generator=g)] + 0.02 * x
    elif kind == 2:  # duplicated / near-duplicated codes
        w[K // 2:] = w[:K - K // 2] * (1 + 1e-4 * ri(0, 3))
    elif kind == 3:  # tiny uniform init
        w = (torch.rand(K, D, device='cuda', generator=g) * 2 - 1) / K
    elif kind == 4:  # heavy-tailed
        x = x * torch.exp(2 * torch.randn(N, 1, device='cuda', generator=g)); w = w * torch.exp(torch.randn(K, 1, device='cuda', generator=g))
    elif kind == 5:  # integer grid (exact ties)
        x = torch.randint(-3, 4, (N, D), device='cuda', generator=g).float(); w = torch.randint(-3, 4, (K, D), device='cuda', generator=g).float()
    elif kind == 6:  # unit-norm codebook (NormalizeCallback): L2 takes the constant-norm form (no bias term in the proposal scores)
        w = torch.nn.functional.normalize(w)
        if ri(0, 2): x = torch.nn.functional.normalize(x)
    x, w = x * scale, w * scale
    if 'VQ_FUZZ_FORCE_EXACT' in os.environ:        # send the first V rows of every batch through the last-resort fp32 pass as well
        from vector_quantization_amd import _lib       # (tuning key 12: both forms of that pass, list lengths around the switch at 16)
        _lib.lib().vqhip_set_tuning(12, [0, 1, 2, 3, 7, 12, 15, 16, 17, 33, 100][ri(0, 11)])
    if os.environ.get('VQ_FUZZ_VERBOSE'): print(f'trial {trials + 1}: N={N} K={K} D={D} {metric} kind={kind} scale={scale}', flush=True)
    xd = x.bfloat16() if (ri(0, 3) == 0 or os.environ.get('VQ_FUZZ_BF16') == '1') else x      # VQ_FUZZ_BF16=1: bf16 latents in every trial
    if metric != 'L2':
        xq = ops.normalize_rows(xd); wq = ops.normalize_rows(w)
        if metric == 'CosineBF16':
            xq, wq = xq.bfloat16().float(), wq.bfloat16().float()
        got = ops.argmin(xq, ops.prepare_codebook(w, metric)) if ri(0, 2) else ops.encode(xd, w, metric)[0]
        ref = ops.argmin_exact(xq, wq, metric)
    else:
        got = ops.argmin(xd, ops.prepare_codebook(w, metric))
        ref = ops.argmin_exact(xd, w, metric)
    nb = int((got != ref).sum().item())
    # NearestAnchor's column argmin (role-swapped pipeline) against the materialised fp32 distance matrix
    if N * K <= 1.5e8 and ri(0, 2) == 0:
        if metric != 'L2':
            d = ops.distance(xq, wq, metric); col = ops.col_argmin(xq, wq, metric)
        else:
            d = ops.distance(xd, w, metric); col = ops.col_argmin(xd, w, metric)
        rows = torch.arange(N, device='cuda')[:, None].expand(N, K)
        cref = torch.where(d == d.min(0, keepdim=True).values, rows, N).min(0).values
        nb += int((col != cref).sum().item())
    # ordered (deterministic) centroid sums against a float64 index_add, and bit-reproducibility
    if K <= 32768 and D % 4 == 0 and ri(0, 8) == 0:
        a = ops.scatter_add_rows(x, ref, K, ordered=True); b = ops.scatter_add_rows(x, ref, K, ordered=True)
        r64 = torch.zeros(K, D, dtype=torch.float64, device='cuda').index_add_(0, ref, x.double())
        tol = 1e-5 * float(x.abs().max()) * max(1.0, N / K) + 1e-30
        if not torch.equal(a, b) or float((a.double() - r64).abs().max()) > tol * 50:
            nb += 1
    trials += 1
    if nb:
        bad += 1
        print(f'MISMATCH trial {trials}: N={N} K={K} D={D} {metric} kind={kind} scale={scale} bf16={xd.dtype} rows={nb}', flush=True)
print(f'{trials} trials, {bad} with mismatches')
sys.exit(1 if bad else 0)
