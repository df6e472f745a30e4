#!/usr/bin/env python3
"""BASELINE configs[1] at small batches (VQGANQuantizer.forward, eval, K = 16384, D = 256, bf16 latents): ms per step for the
tuning states given as key=value pairs on the command line, in one process, alternating.
usage: ab_small_batch.py images  [6=1] [2=8] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vector_quantization_amd import _lib
images = int(sys.argv[1]) if len(sys.argv) > 1 else 32
states = [None] + [tuple(int(v) for v in a.split('=')) for a in sys.argv[2:]]
L = _lib.lib()
K, D, dev = 16384, 256, torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.randn(K, D, device=dev, generator=g)
x = torch.randn(images * 256, D, device=dev, generator=g).bfloat16()
q = bench.build_module(bench.quantizer_cfg(K, D, 'L2'), dev, w, train=False)
DEFAULTS = {6: 2, 2: 0}
def run(n):
    with torch.no_grad():
        for _ in range(n):
            out = q(x, {})
    return out
res = {s: [] for s in states}
ref = None
for rnd in range(5):
    for s in states:
        for k, v in DEFAULTS.items():
            L.vqhip_set_tuning(k, v)
        if s is not None:
            L.vqhip_set_tuning(*s)
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = run(200); torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t0) / 200 * 1e3)
        idx = out[2]['quant']
        ref = idx if ref is None else ref
        assert torch.equal(idx, ref)
for s in states:
    v = sorted(res[s])
    print(f'{images} images, tuning {s}: median {v[len(v) // 2]:.4f} ms  min {v[0]:.4f}  -> {images * 256 / v[len(v) // 2] / 1e3:.1f} M tokens/s')
