"""The real update callbacks at world size 2 on the GPU.

Two fresh processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the product stages the
collectives gloo cannot run on device tensors through the host: utils.all_gather / gather_to_rank0).  Each rank runs the
nn.Module path — VQKDCallback / CVQVAECallback train steps, VQKDCallback.lazy_init_weights with its gather + broadcast,
NearestAnchor with sync=False (anchors averaged) and sync=True (global column argmin) — on rows rank::2.  The results
are compared with the fixtures a two-process run of the REFERENCE's own callbacks produced (oracle/make_golden.py:
_rank_worker), and DRY_RUN=1 arms the reference's is_sync invariant (callbacks/update.py:54-55, anchors.py:52-53,62-63)
inside the product code.
"""
import json
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
EMB = 'torch_nn_modules_sparse_Embedding'


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _cfg(kind, K, D, dist='Cosine', sync=False):
    emb = dict(type=EMB, num_embeddings=K, embedding_dim=D)
    if kind == 'vqkd':       # configs/vqkd/model.py:20-26
        return dict(type='VQKDQuantizer', embedding=emb, distance=dict(type='CosineDistance'),
                    callbacks=[dict(type='VQKDCallback', ema=dict())],
                    losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))
    return dict(type='VQGANQuantizer', embedding=emb, distance=dict(type=f'{dist}Distance'),      # configs/cvqvae/quantizer.py
                callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor', sync=sync))],
                losses=dict(vqgan_loss=dict(type='VQGANLoss')))


def _worker(rank, world, port, outdir):
    import random

    import torch
    import torch.distributed as dist

    from oracle import synth
    from vector_quantization_amd import Config, build_quantizer
    from vector_quantization_amd.utils import is_sync

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DRY_RUN='1')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    spec = json.loads(str(np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    xr = torch.from_numpy(x[rank::world]).cuda()
    rec = {}

    def build(cfg, init=None, weight=None):
        torch.manual_seed(0)                       # identical nn.Embedding init on every rank (DDP's broadcast)
        q = build_quantizer(cfg)
        q.train()
        q.init_weights(Config(init or {}))
        q = q.cuda()
        if weight is not None:
            q._forward_pre_hooks.clear()           # start from the given codebook: no lazy k-means init
            with torch.no_grad():
                q.embedding.weight.copy_(torch.from_numpy(weight))
        return q

    q = build(_cfg('vqkd', K, D), weight=w)
    z, loss, memo = q(xr, {})
    assert is_sync(q.embedding.weight.detach())
    rec['vqkd_quant'], rec['vqkd_w_new'] = memo['quant'].cpu().numpy(), q.embedding.weight.detach().cpu().numpy()
    for dname in ('L2', 'Cosine'):
        for sync in (False, True):
            q = build(_cfg('cvq', K, D, dname, sync), init=dict(type='vqgan'), weight=w)
            z, loss, memo = q(xr, {})
            assert is_sync(q.embedding.weight.detach()) and is_sync(q.get_buffer('_probability'))
            tag = f'cvq_{dname.lower()}_{"sync" if sync else "avg"}'
            rec[f'{tag}_w_new'] = q.embedding.weight.detach().cpu().numpy()
            rec[f'{tag}_p'] = q.get_buffer('_probability').cpu().numpy()
            rec[f'{tag}_quant'] = memo['quant'].cpu().numpy()
    # several CVQ-VAE steps with the packed sparse exchange (default) against the reference's dense data flow: bit-identical
    # codebooks, ONE collective per step, and an exchange that shrinks with the codes that are in regular use
    from vector_quantization_amd.utils import exchange_log
    gen = synth.rng(100)
    wk = synth.unit_rows(gen.standard_normal((K, D), dtype=np.float32))
    steps = [(gen.standard_normal((N, D), dtype=np.float32) * np.float32(0.3) + wk[gen.integers(0, K // 8, N)])[rank::world]
             for _ in range(5)]
    runs = {}
    for sparse in (None, False):
        cfg = _cfg('cvq', K, D, 'Cosine', False)
        cfg['callbacks'][0]['sparse_anchors'] = sparse
        q = build(cfg, init=dict(type='vqgan'), weight=wk)
        calls, nbytes, rows = [], [], []
        for xs in steps:
            exchange_log.start()
            q(torch.from_numpy(xs).cuda(), {})
            st = exchange_log.stop()
            calls.append(st['calls']); nbytes.append(st['bytes']); rows.append(q._callbacks.callbacks[0].last_exchange_rows)
        assert is_sync(q.embedding.weight.detach())
        runs[sparse] = (q.embedding.weight.detach().clone(), q.get_buffer('_probability').clone(), calls, nbytes, rows)
    assert torch.equal(runs[None][0], runs[False][0]) and torch.equal(runs[None][1], runs[False][1])
    assert runs[None][2] == [1] * 5 and runs[False][2] == [2] * 5, (runs[None][2], runs[False][2])
    assert runs[None][3] == [4 * (2 * K + 4 + m * D) for m in runs[None][4]]
    assert runs[None][4][0] == K and runs[None][4][-1] < K, runs[None][4]       # first step: p = 0, every code listed
    rec['sparse_rows'] = np.asarray(runs[None][4])
    rec['sparse_w'] = runs[None][0].cpu().numpy()
    # lazy k-means init: gather to rank 0, Lloyd iterations there, broadcast (callbacks.py:77-112).  DRY_RUN off, as in
    # the reference run (rank 0 alone calls _update_embedding inside the loop)
    os.environ['DRY_RUN'] = ''
    random.seed(1234)
    q = build(_cfg('vqkd', 64, D))
    q(xr[:256], {})
    assert is_sync(q.embedding.weight.detach()) and len(q._forward_pre_hooks) == 0
    rec['lazy_w'] = q.embedding.weight.detach().cpu().numpy()
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **rec)
    dist.barrier()
    dist.destroy_process_group()


def test_callbacks_at_world_size_two(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))) for r in range(2))
    np.testing.assert_array_equal(r0['sparse_rows'], r1['sparse_rows'])                  # every rank sized the same exchange
    for k in r0:
        if k.endswith('w_new') or k.endswith('_p') or k in ('lazy_w', 'sparse_w'):
            assert r0[k].tobytes() == r1[k].tobytes(), f'ranks disagree on {k}'          # bit-identical codebooks
    g = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    np.testing.assert_array_equal(r0['vqkd_quant'], g['quant_rank0'].astype(np.int64))
    np.testing.assert_array_equal(r1['vqkd_quant'], g['quant_rank1'].astype(np.int64))
    np.testing.assert_allclose(r0['vqkd_w_new'], g['w_new_2rank'], rtol=0, atol=3e-6)
    for dname in ('l2', 'cosine'):
        g = np.load(os.path.join(GOLDEN, f'update_cvq_{dname}.npz'))
        np.testing.assert_allclose(r0[f'cvq_{dname}_avg_p'], g['p_2rank'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(r0[f'cvq_{dname}_avg_w_new'], g['w_new_2rank'], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r0[f'cvq_{dname}_sync_p'], g['p_2rank_sync'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(r0[f'cvq_{dname}_sync_w_new'], g['w_new_2rank_sync'], rtol=0, atol=3e-6)
    g = np.load(os.path.join(GOLDEN, 'lazy_init_2rank.npz'))
    np.testing.assert_allclose(r0['lazy_w'], g['w'], rtol=0, atol=1e-5)
