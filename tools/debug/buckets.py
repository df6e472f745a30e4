"""GraphedQuantizer capacity buckets: does an 8192 bucket help the CVQ-VAE per-rank step (7 311 codes listed)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_train_shapes as B
from vector_quantization_amd import affinity
from vector_quantization_amd.graphs import GraphedQuantizer
torch.cuda.init(); affinity.bind_rank(0, 0, probe=True)
for name in ('cvq', 'cluster'):
    N, K, D, cfg = B.SHAPES[name]
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(3407)
    w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
    pool = [(w[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.05 * torch.randn(N, D, device=dev, generator=g)).requires_grad_(True) for _ in range(21)]
    gz = torch.randn(N, D, device=dev, generator=g) / (N * D)
    for caps in ((256, 4096), (128, 1024, 4096, 8192)):
        q = B.build(cfg, K, D, w, dev)
        params = [p for p in q.parameters() if p.requires_grad]
        turn = [0]
        def step(call):
            xin = pool[turn[0] % 21]; turn[0] += 1
            for p in params: p.grad = None
            xin.grad = None
            z, loss = call(xin)[:2]
            torch.autograd.backward([loss, z], [None, gz])
        for _ in range(150): step(lambda xin: q(xin, {}))
        gq = GraphedQuantizer(q, pool[0].detach(), bucket_caps=caps)
        for _ in range(50): step(gq)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): step(gq)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
        print(f'{name} buckets {caps}: {best:.4f} ms per replayed step; listed {q._callbacks.callbacks[0].last_exchange_rows}', flush=True)
