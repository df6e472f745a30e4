#!/usr/bin/env python3
"""Host time of the pieces of an eager CVQ-VAE forward (no profiler: time.perf_counter around 2000 calls of each piece)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_train_shapes as B
from vector_quantization_amd import affinity
affinity.bind_rank(0, 0, probe=True)
name = sys.argv[1] if len(sys.argv) > 1 else 'cvq'
N, K, D, cfg = B.SHAPES[name]
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
x = (w[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.05 * torch.randn(N, D, device=dev, generator=g)).requires_grad_(True)
gz = torch.randn(N, D, device=dev, generator=g) / (N * D)
q = B.build(cfg, K, D, w, dev)
for _ in range(100):
    z, loss, memo = q(x, {})
torch.cuda.synchronize()
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
cb = q._callbacks.callbacks[0] if q._callbacks.callbacks else None
print(f'{name}: forward (host, no backward)            {t(lambda: q(x, {}), 500):7.1f} us')
print(f'{name}: _one_call_step(x)                       {t(lambda: q._one_call_step(x)):7.1f} us')
if cb is not None and hasattr(cb, 'fused_forward_ok'):
    print(f'{name}: fused_forward_ok(x)                     {t(lambda: cb.fused_forward_ok(x)):7.1f} us')
print(f'{name}: six torch.empty                         {t(lambda: [torch.empty(N, dtype=torch.int64, device=dev), torch.empty(K, dtype=torch.int32, device=dev), torch.empty(N, D, device=dev), torch.empty(N, D, device=dev), torch.empty(4, device=dev), torch.empty(1 << 20, dtype=torch.uint8, device=dev)]):7.1f} us')
from vector_quantization_amd import _lib
def mk():
    a = _lib.CvqForwardArgs()
    for f, _ in a._fields_[:40]:
        setattr(a, f, 1)
print(f'{name}: a ctypes argument block filled          {t(mk):7.1f} us')
from vector_quantization_amd.utils import exchanging
print(f'{name}: exchanging()                            {t(exchanging):7.1f} us')
def imp():
    from vector_quantization_amd import functional as VF, train_step
    from vector_quantization_amd.utils import all_reduce_min, all_reduce_sum, get_rank
print(f'{name}: three function-level imports            {t(imp):7.1f} us')
z, loss, memo = q(x, {})
print(f'{name}: backward (host)                         {t(lambda: torch.autograd.backward([q(x, {})[1]], [None]), 300) - t(lambda: q(x, {}), 300):7.1f} us')
