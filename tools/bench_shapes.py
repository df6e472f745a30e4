#!/usr/bin/env python3
"""Throughput of the encode path on the other BASELINE.json shapes (C3 VQ-KD cosine D=32, LlamaGen D=8, C1) plus the
training-step pieces (VQ-KD / CVQ updates).  Reports tokens/s per shape; one process, CUDA events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops

def timeit(fn, reps=20, warm=3):
    # (a first untimed pass of the same length: the first time a process has this many launches in flight the HIP runtime
    #  grows its pools and the HOST spends ~2 ms per call for a few dozen calls — seen as a 4x slower quantize() on whichever
    #  configuration came first, tools history in profiles/r03_shapes.txt)
    for _ in range(max(warm, reps)): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

g = torch.Generator(device='cuda').manual_seed(3407)
def shape(name, N, K, D, metric, normalize=False, dtype=torch.float32):
    w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g).to(dtype)
    def enc():
        ww, xx = w, x
        if normalize: ww, xx = ops.normalize_rows(w), ops.normalize_rows(x)
        cb = ops.prepare_codebook(ww, metric)
        xq = ops.normalize_rows(xx) if metric == 'Cosine' else xx
        return ops.argmin(xq, cb, return_stats=True)
    def enc1():             # the same work as ONE library call (vqhip_encode: two or three launches less)
        ww, xx = w, x
        if normalize: ww, xx = ops.normalize_rows(w), ops.normalize_rows(x)
        return ops.encode(xx, ww, metric)[0]
    idx, st = enc(); t = timeit(lambda: enc())
    assert torch.equal(enc1(), idx)
    t1 = timeit(lambda: enc1())
    print(f'{name:34s} N={N:7d} K={K:5d} D={D:3d} {metric:6s}: {t*1e3:8.3f} ms  {N/t/1e6:8.1f} Mtok/s  '
          f'one call {t1*1e3:8.3f} ms {N/t1/1e6:8.1f} Mtok/s  rescan={int(st[0])} multi={int(st[1])} exact={int(st[2])}', flush=True)

shape('C1 VQGAN small', 1024, 1024, 256, 'L2')
shape('C2 VQGAN 32 img bf16', 8192, 16384, 256, 'L2', dtype=torch.bfloat16)
shape('C2 VQGAN 256 img bf16', 65536, 16384, 256, 'L2', dtype=torch.bfloat16)
shape('C2 VQGAN 256 img fp32 x', 65536, 16384, 256, 'L2')
shape('C3 VQ-KD cosine', 100352, 8192, 32, 'Cosine')
shape('C4 CVQ cosine per-rank', 3072, 16384, 256, 'Cosine')
shape('C5 LlamaGen D=8 normalize+L2', 65536, 16384, 8, 'L2', normalize=True)
shape('cluster D=768 cosine', 8192, 8192, 768, 'Cosine')
shape('cluster D=768 cosine, 64k rows', 65536, 8192, 768, 'Cosine')
shape('D=1024 L2', 65536, 8192, 1024, 'L2')
shape('D=1032 (fp32 route)', 8192, 8192, 1032, 'L2')
# training-step pieces (per-rank C4 shape)
N, K, D = 3072, 16384, 256
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
t = timeit(lambda: ops.col_argmin(x, w, 'L2'), reps=5); print(f'col_argmin (NearestAnchor) N={N} K={K} D={D}: {t*1e3:.3f} ms')
idx = ops.argmin(x, ops.prepare_codebook(w, 'L2'))
t = timeit(lambda: ops.scatter_add_rows(x, idx, K)); print(f'scatter_add_rows: {t*1e3:.3f} ms')
t = timeit(lambda: ops.hist(idx, K)); print(f'hist: {t*1e3:.3f} ms')

# model-level quantize(): NCHW latent map (rearrangements folded into the encode / gather kernels) vs channels-last (views)
from vector_quantization_amd import build_quantizer, Config, tokenization as T
B, C, H, W, K = 256, 256, 16, 16, 16384
q = build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=C),
                         distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
q.init_weights(Config(type='vqgan')); q = q.cuda().eval()
with torch.no_grad():
    q.embedding.weight.copy_(torch.randn(K, C, device='cuda', generator=g))
    xm = torch.randn(B, C, H, W, device='cuda', generator=g).bfloat16()
    for name, fmt in (('NCHW', torch.contiguous_format), ('channels-last', torch.channels_last)):
        xi = xm.contiguous(memory_format=fmt)
        t = timeit(lambda: T.quantize(q, xi, {}), warm=100)     # (the host-side pool growth described in timeit lasts a few dozen calls)
        print(f'quantize() {name:14s} B={B} {C}x{H}x{W} K={K}: {t*1e3:.3f} ms  {B*H*W/t/1e6:.1f} Mtok/s')
# the same in train mode with the backward from a given upstream gradient of the map (what the decoder would hand back)
q.train()
gmap = torch.randn(B, C, H, W, device='cuda', generator=g) / (B * C * H * W)
for name, fmt in (('NCHW', torch.contiguous_format), ('channels-last', torch.channels_last)):
    xi = xm.contiguous(memory_format=fmt).requires_grad_(True)
    gi = gmap.contiguous(memory_format=fmt)
    def fb():
        xi.grad = None
        q.embedding.weight.grad = None
        z, loss, _ = T.quantize(q, xi, {})
        torch.autograd.backward([z, loss], [gi, None])
    t = timeit(fb, warm=100)
    print(f'quantize() fwd+bwd {name:14s} B={B} {C}x{H}x{W} K={K}: {t*1e3:.3f} ms  {B*H*W/t/1e6:.1f} Mtok/s')
