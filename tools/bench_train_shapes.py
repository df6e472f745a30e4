#!/usr/bin/env python3
"""Training steps of the drop-in nn.Module at the per-rank shapes of the reference's shipped training configs (8 ranks):

  cvq      3 072 x 16384 x 256  cosine, CVQVAECallback(NearestAnchor), VQGANLoss       configs/vqgan/interface.py:8 (96 images / 8), configs/cvqvae/quantizer.py
  vqkd    12 544 x  8192 x  32  cosine, VQKDCallback(ema), CommitmentLoss(norm=True)   configs/vqkd/interface.py:8 (512 / 8 x 196), configs/vqkd/model.py:20-26
  cluster  6 272 x  8192 x 768  cosine, CVQVAECallback(NearestAnchor sync), CodebookLoss   configs/cluster/interface.py:8 (256 / 8 x 196), configs/cluster/model.py
  llamagen 4 096 x 16384 x   8  L2 + NormalizeCallback, VQGANLoss                       configs/llamagen/vqgan.py:8-20 (128 / 8 x 256)

One step = forward (encode, codebook update of the callback, decode, loss) + backward from a given upstream gradient of
the straight-through output, on fresh latents every step (a pool drawn around the codebook rows, as bench.py's cvq block).
Per shape: eager ms per step back to back, the time after which the HOST has issued a step (host-bound when the two
agree), and the same step replayed from HIP graphs (GraphedQuantizer).  usage: bench_train_shapes.py [names...] [--bf16] [--no-bind]
VQ_TRAIN_STEPS / VQ_TRAIN_SETTLE / VQ_TRAIN_POOL override the step counts and the number of pooled batches; VQ_TRAIN_NO_GRAPH=1 skips
the graphed leg."""
import functools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

print = functools.partial(print, flush=True)
EMB = 'torch_nn_modules_sparse_Embedding'


def emb(K, D):
    return dict(type=EMB, num_embeddings=K, embedding_dim=D)


SHAPES = {
    'cvq': (3072, 16384, 256, dict(type='VQGANQuantizer', distance=dict(type='CosineDistance'),
                                   losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                                   callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])),
    'vqkd': (12544, 8192, 32, dict(type='VQKDQuantizer', distance=dict(type='CosineDistance'),
                                   losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))),
                                   callbacks=[dict(type='VQKDCallback', ema=dict())])),
    'cluster': (6272, 8192, 768, dict(type='VQGANQuantizer', distance=dict(type='CosineDistance'),
                                      losses=dict(vqgan_loss=dict(type='CodebookLoss')),
                                      callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor', sync=True))])),
    'llamagen': (4096, 16384, 8, dict(type='VQGANQuantizer', distance=dict(type='L2Distance'),
                                      losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                                      callbacks=[dict(type='NormalizeCallback')])),
    # the same CVQ-VAE step at the bulk per-rank batch of BASELINE configs[3]'s second size
    'cvq64k': (65536, 16384, 256, dict(type='VQGANQuantizer', distance=dict(type='CosineDistance'),
                                       losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                                       callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])),
}


def build(cfg, K, D, w, dev):
    from vector_quantization_amd import Config, build_quantizer
    q = build_quantizer(dict(cfg, embedding=emb(K, D)))
    q.train(True)
    q.init_weights(Config(type='vqgan') if cfg['type'] == 'VQGANQuantizer' else Config())
    q = q.to(dev)
    q._forward_pre_hooks.clear()          # VQ-KD: the k-means lazy init is not part of a steady-state step
    with torch.no_grad():
        q.embedding.weight.copy_(w)
    if cfg['type'] == 'VQKDQuantizer':    # configs/vqkd/model.py:76-82: no_grad on the quantizer's parameters
        for p in q.parameters():
            p.requires_grad_(False)
    return q


def run(name, bf16):
    N, K, D, cfg = SHAPES[name]
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(3407)
    w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g))
    npool = int(os.environ.get('VQ_TRAIN_POOL', '0')) or max(8, min(64, (1 << 24) // (N * D)))
    pool = []
    for _ in range(npool):
        t = w[torch.randint(0, K, (N,), device=dev, generator=g)] + 0.05 * torch.randn(N, D, device=dev, generator=g)
        if bf16:
            t = t.bfloat16()
        pool.append(t.requires_grad_(True))
    gz = torch.randn(N, D, device=dev, generator=g) / (N * D)
    steps = int(os.environ.get('VQ_TRAIN_STEPS', '200'))
    settle = int(os.environ.get('VQ_TRAIN_SETTLE', '150'))

    def make_step(call, params):
        turn = [0]

        def step():
            xin = pool[turn[0] % npool]
            turn[0] += 1
            for p_ in params:
                p_.grad = None
            xin.grad = None
            z, loss = call(xin)
            torch.autograd.backward([loss, z], [None, gz])
        return step

    def timed(step, label):
        for _ in range(30):
            step()
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rec = ((t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3)
            if best is None or rec[0] < best[0]:
                best = rec
        print(f'{name:9s} {label:8s} N={N:6d} K={K:5d} D={D:3d} {"bf16" if bf16 else "fp32"}: {best[0]:7.4f} ms per step '
              f'({N / best[0] / 1e3:7.2f} Mtok/s), host has issued a step after {best[1]:7.4f} ms')
        return best

    q = build(cfg, K, D, w, dev)
    params = [p for p in q.parameters() if p.requires_grad]
    eager = make_step(lambda xin: q(xin, {})[:2], params)
    for _ in range(settle):
        eager()
    timed(eager, 'eager')
    cb0 = q._callbacks.callbacks[0] if len(q._callbacks.callbacks) else None
    if getattr(cb0, 'last_exchange_rows', None) is not None:
        print(f'{name:9s}          codes listed for an anchor in the last step: {cb0.last_exchange_rows} of {K}')
    if os.environ.get('VQ_TRAIN_NO_GRAPH') == '1':
        return
    from vector_quantization_amd.graphs import GraphedQuantizer
    qg = build(cfg, K, D, w, dev)
    qg.load_state_dict(q.state_dict())
    try:
        gq = GraphedQuantizer(qg, pool[0].detach())
    except Exception as exc:                                  # noqa: BLE001
        print(f'{name:9s} graphed : capture failed: {type(exc).__name__}: {exc}')
        return
    gparams = [p for p in qg.parameters() if p.requires_grad]
    timed(make_step(lambda xin: gq(xin)[:2], gparams), 'graphed')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    bf16 = '--bf16' in sys.argv
    if '--no-bind' not in sys.argv:       # these steps are as much host work as GPU work: one L3 slice of the GPU's NUMA node (affinity.py)
        from vector_quantization_amd import affinity
        torch.cuda.init()
        info = affinity.bind_rank(0, 0, probe=True)
        print(f'host threads bound to CPUs {info["cpus"]} (GPU-local NUMA node: {info["numa_local"]})' if info else 'host threads: not bound')
    for name in (args or ['cvq', 'vqkd', 'cluster', 'llamagen']):
        run(name, bf16)
