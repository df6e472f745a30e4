#!/usr/bin/env python3
"""NearestAnchor(sync=True) across ranks (configs/cluster/model.py:28: 6 272 x 8192 x 768 per rank, cosine): the key exchange
(the default) against the reference's data flow (latents all-gathered, column argmin over world x N tokens on
every rank: `sparse_anchors=False`), at world sizes 1, 2, 4, 8 with every rank on the one GPU a builder box has (gloo).

What a shared GPU can and cannot show: the ranks' kernels time-share one device, so the WALL time of a step grows with the
world whatever the route — what is compared is (a) the GPU time one rank's step needs (`gpu_ms`: HIP events around the step on
the rank's stream, minimum over the timed steps: the least-contended one), (b) the bytes a rank hands to the collectives per
step, (c) the wall time per step with all ranks running (host-staged gloo collectives included — an upper bound for both).
usage: bench_sync_ranks.py [worlds...]   (default 1 2 8)"""
import functools
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

print = functools.partial(print, flush=True)
EMB = 'torch_nn_modules_sparse_Embedding'
N, K, D = 6272, 8192, 768


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist

    from vector_quantization_amd import Config, build_quantizer
    from vector_quantization_amd.utils import exchange_log, is_sync
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(3407)
    w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    pool = [(w[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.05 * torch.randn(N, D, device=dev, generator=g)).requires_grad_(True)
            for _ in range(8)]
    gz = torch.randn(N, D, device=dev, generator=g) / (N * D)
    steps, settle = int(os.environ.get('VQ_SYNC_STEPS', '40')), int(os.environ.get('VQ_SYNC_SETTLE', '120'))
    out = {}
    for route in ('keys', 'gather'):
        cfg = dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D), distance=dict(type='CosineDistance'),
                   losses=dict(vqgan_loss=dict(type='CodebookLoss')),
                   callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor', sync=True),
                                   sparse_anchors=None if route == 'keys' else False)])
        q = build_quantizer(cfg)
        q.train(True)
        q.init_weights(Config(type='vqgan'))
        q = q.to(dev)
        with torch.no_grad():
            q.embedding.weight.copy_(w)
        params = [p for p in q.parameters() if p.requires_grad]
        turn = [0]

        def step():
            x = pool[turn[0] % len(pool)]
            turn[0] += 1
            for p in params:
                p.grad = None
            x.grad = None
            z, loss, memo = q(x, {})
            torch.autograd.backward([loss, z], [None, gz])

        for _ in range(settle if route == 'keys' else max(10, settle // 6)):      # (the gather route is slow: fewer settle steps)
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        gpu = []
        t0 = time.perf_counter()
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            step()
            b.record()
            gpu.append((a, b))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ev = sorted(a.elapsed_time(b) for a, b in gpu)
        exchange_log.start()
        for _ in range(5):
            step()
        st = exchange_log.stop()
        cb = q._callbacks.callbacks[0]
        out[route] = dict(wall_ms_per_step=wall, gpu_ms_min=ev[0], gpu_ms_median=ev[len(ev) // 2], collectives_per_step=st['calls'] / 5,
                          logged_exchange_bytes_per_step=st['bytes'] / 5, listed_codes=cb.last_exchange_rows,
                          in_sync=bool(is_sync(q.embedding.weight.detach())) if world > 1 else True)
    # the gather route's all_gather of latents / tokens / probabilities is not a logged collective: its size by construction
    out['gather']['all_gather_bytes_per_rank_per_step'] = world * (N * D * 4 + N * 8 + K * 4) if world > 1 else 0
    if rank == 0:
        json.dump(out, open(os.path.join(outdir, f'ws{world}.json'), 'w'))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    import tempfile

    import torch.multiprocessing as mp
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 8]
    print(f'NearestAnchor(sync=True), {N} x {K} x {D} cosine per rank, every rank on cuda:0 over gloo')
    print(f'{"world":>5} {"route":>7} {"gpu ms (min)":>13} {"gpu ms (med)":>13} {"wall ms":>9} {"coll/step":>9} {"bytes/step":>12} {"listed":>7} in_sync')
    with tempfile.TemporaryDirectory() as td:
        for world in worlds:
            mp.spawn(worker, args=(world, _free_port(), td), nprocs=world, join=True)
            rec = json.load(open(os.path.join(td, f'ws{world}.json')))
            for route in ('keys', 'gather'):
                r = rec[route]
                nbytes = r['logged_exchange_bytes_per_step'] + r.get('all_gather_bytes_per_rank_per_step', 0)
                print(f'{world:>5} {route:>7} {r["gpu_ms_min"]:>13.3f} {r["gpu_ms_median"]:>13.3f} {r["wall_ms_per_step"]:>9.3f} '
                      f'{r["collectives_per_step"]:>9.1f} {nbytes:>12.0f} {str(r["listed_codes"]):>7} {r["in_sync"]}')


if __name__ == '__main__':
    main()
