#!/usr/bin/env python3
"""Is the graph-replayed CVQ-VAE step (bench.py --workload cvq, `ms_per_step_graphed`) bound by the host?  Times the loop of
replayed steps twice: up to the point where the host has ISSUED all of them (no synchronisation) and to the end of the GPU
work.  usage: cvq_host_bound.py [tokens] [steps]"""
import os, sys, time
import functools
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vector_quantization_amd.graphs import GraphedQuantizer

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K, D, dev = 16384, 256, torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
pool = [(w[torch.randint(0, K, (tokens,), device=dev, generator=g)] + 0.05 * torch.randn(tokens, D, device=dev, generator=g))
        .requires_grad_(True) for _ in range(64)]
gz = torch.randn(tokens, D, device=dev, generator=g) / (tokens * D)
cb_cfg = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]
q = bench.build_module(bench.quantizer_cfg(K, D, 'Cosine', cb_cfg), dev, w, train=True)
for i in range(150):
    xin = pool[i % 64]
    xin.grad = None
    for p_ in q.parameters():
        p_.grad = None
    z, loss, _ = q(xin, {})
    torch.autograd.backward([loss, z], [None, gz])
    del z, loss
qg = bench.build_module(bench.quantizer_cfg(K, D, 'Cosine', cb_cfg), dev, w, train=True)
qg.load_state_dict(q.state_dict())
print('settled')
gq = GraphedQuantizer(qg, pool[0].detach())
print('captured')
params = list(qg.parameters())
turn = [0]

def step():
    xin = pool[turn[0] % 64]
    turn[0] += 1
    for p_ in params:
        p_.grad = None
    xin.grad = None
    z, loss, _ = gq(xin)
    torch.autograd.backward([loss, z], [None, gz])

for _ in range(30):
    step()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{steps} replayed steps: issued after {(t1 - t0) / steps * 1e6:.1f} us per step, done after {(t2 - t0) / steps * 1e6:.1f} us per step')
