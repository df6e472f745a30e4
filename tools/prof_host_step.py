#!/usr/bin/env python3
"""Host-side profile (cProfile) of an eager training step at a per-rank shape (tools/bench_train_shapes.py names): where the
Python time between launches goes.  usage: prof_host_step.py cvq|vqkd|cluster|llamagen"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_train_shapes as B

name = sys.argv[1] if len(sys.argv) > 1 else 'vqkd'
N, K, D, cfg = B.SHAPES[name]
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
pool = [(w[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.05 * torch.randn(N, D, device=dev, generator=g)).requires_grad_(True) for _ in range(16)]
gz = torch.randn(N, D, device=dev, generator=g) / (N * D)
q = B.build(cfg, K, D, w, dev)
params = [p for p in q.parameters() if p.requires_grad]
turn = [0]

def step():
    xin = pool[turn[0] % 16]; turn[0] += 1
    for p in params: p.grad = None
    xin.grad = None
    z, loss, _ = q(xin, {})
    torch.autograd.backward([loss, z], [None, gz])

for _ in range(200): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'{name}: {(t2 - t0) / 300 * 1e3:.4f} ms per step, issued after {(t1 - t0) / 300 * 1e3:.4f}')
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
