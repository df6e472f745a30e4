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
    full = [gen.standard_normal((N, D), dtype=np.float32) * np.float32(0.3) + wk[gen.integers(0, K // 8, N)] for _ in range(5)]
    steps = [f[rank::world] for f in full]
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
    assert torch.equal(runs[None][1], runs[False][1])                          # probabilities: integer counts, exact either way
    if world <= 2:
        assert torch.equal(runs[None][0], runs[False][0])
    else:
        # more than two ranks: a ring all-reduce adds the ranks' contributions in an order that depends on WHERE in the buffer an
        # element sits, and an anchor row sits at another offset in the packed buffer than in the dense [K, D] tensor — both sums are
        # the reference's sum (anchors.py:65-67) to the last bits, neither is "the" order
        torch.testing.assert_close(runs[None][0], runs[False][0], rtol=0, atol=1e-6)
    assert runs[None][2] == [1] * 5 and runs[False][2] == [2] * 5, (runs[None][2], runs[False][2])
    assert runs[None][3] == [4 * (2 * K + 4 + m * D) for m in runs[None][4]]
    assert runs[None][4][0] == K and runs[None][4][-1] < K, runs[None][4]       # first step: p = 0, every code listed
    rec['sparse_rows'] = np.asarray(runs[None][4])
    rec['sparse_w'] = runs[None][0].cpu().numpy()
    # NearestAnchor(sync=True) — the cluster config (configs/cluster/model.py:28): the key exchange of SURVEY.md §8e (default; one
    # call per forward and hook by hook) against the reference's data flow (latents gathered, column argmin over ALL of them on
    # every rank: sparse_anchors=False).  One rank contributes each anchor row, so the codebooks are bit-identical at ANY world
    # size; the exchange is a MIN all-reduce of 8 M bytes of keys and the packed SUM of 4 (2K + 4 + M D) bytes — no latent
    # travels.  The duplicated latents of `ties` make two ranks hold the SAME nearest row of a code: the lowest (rank, row) wins
    ties = [s.copy() for s in steps]
    for s in ties:
        s[1::2] = full[0][:1]                      # every rank's odd rows = one and the same latent
    for dname in ('Cosine', 'L2'):
        sruns = {}
        for route in ('one_call', 'hooks', 'gather'):
            cfg = _cfg('cvq', K, D, dname, True)
            cfg['callbacks'][0]['sparse_anchors'] = False if route == 'gather' else None
            q = build(cfg, init=dict(type='vqgan'), weight=wk)
            q.one_call_steps = route == 'one_call'
            calls, nbytes, rows = [], [], []
            for xs in steps + ties:
                exchange_log.start()
                q(torch.from_numpy(xs).cuda(), {})
                st = exchange_log.stop()
                calls.append(st['calls']); nbytes.append(st['bytes']); rows.append(q._callbacks.callbacks[0].last_exchange_rows)
            assert is_sync(q.embedding.weight.detach()) and is_sync(q.get_buffer('_probability'))
            sruns[route] = (q.embedding.weight.detach().clone(), q.get_buffer('_probability').clone(), calls, nbytes, rows)
        for route in ('one_call', 'hooks'):
            assert torch.equal(sruns[route][1], sruns['gather'][1]), (dname, route)
            assert torch.equal(sruns[route][0], sruns['gather'][0]), (dname, route)
            assert sruns[route][2] == [2 if m else 1 for m in sruns[route][4]], (dname, route, sruns[route][2])
            assert sruns[route][3] == [8 * m + 4 * (2 * K + 4 + m * D) for m in sruns[route][4]], (dname, route)
        assert sruns['one_call'][4] == sruns['hooks'][4] and sruns['one_call'][4][0] == K
        rec[f'sync_rows_{dname.lower()}'] = np.asarray(sruns['one_call'][4])
        rec[f'sync_w_{dname.lower()}'] = sruns['one_call'][0].cpu().numpy()
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


def test_callbacks_at_world_size_eight(tmp_path):
    """Eight ranks without eight GPUs: eight processes on the shared cuda:0 over gloo run the nn.Module path on rows r::8 —
    VQ-KD step, CVQ-VAE steps with NearestAnchor sync False / True (L2 and cosine), the packed sparse exchange against the dense
    flow over several steps (exact counts: ONE collective of 4 (2K + 4 + M D) bytes per step), the k-means lazy init with its
    gather and broadcast — with DRY_RUN=1 arming the reference's is_sync asserts.  Compared with an EIGHT-process run of the
    reference's own callbacks (tests/golden/update_8rank.npz, oracle/make_golden.py: eight_rank_reference)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    ranks = [dict(np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))) for r in range(8)]
    for r in ranks[1:]:
        np.testing.assert_array_equal(ranks[0]['sparse_rows'], r['sparse_rows'])
        for k in ranks[0]:
            if k.endswith('w_new') or k.endswith('_p') or k in ('lazy_w', 'sparse_w') or k.startswith('sync_w'):
                assert ranks[0][k].tobytes() == r[k].tobytes(), f'ranks disagree on {k}'
    g = np.load(os.path.join(GOLDEN, 'update_8rank.npz'))
    for r in range(8):
        np.testing.assert_array_equal(ranks[r]['vqkd_quant'], g['vqkd_quant'][r].astype(np.int64))
    np.testing.assert_allclose(ranks[0]['vqkd_w_new'], g['vqkd_w_new'], rtol=0, atol=3e-6)
    for tag in ('cvq_l2_avg', 'cvq_cosine_avg', 'cvq_l2_sync', 'cvq_cosine_sync'):
        for r in range(8):
            np.testing.assert_array_equal(ranks[r][f'{tag}_quant'], g[f'{tag}_quant'][r].astype(np.int64))
        np.testing.assert_allclose(ranks[0][f'{tag}_p'], g[f'{tag}_p'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(ranks[0][f'{tag}_w_new'], g[f'{tag}_w_new'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(ranks[0]['lazy_w'], g['lazy_w'], rtol=0, atol=1e-5)


def test_bench_eight_ranks_on_a_shared_gpu(tmp_path):
    """`python bench.py --gpus 8` to completion with every rank on cuda:0 over gloo (VQ_BENCH_SHARE_GPU=1): the only rehearsal
    of the driver's 8-GPU command a one-GPU box allows — the launcher, the rendezvous, the barriers and max-reduction of the
    timing, the communicating cvq block with its exchange accounting and the codebook-in-sync check."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQ_BENCH_SHARE_GPU='1', VQ_BENCH_CVQ_SETTLE='30')
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--images', '16',
                          '--min-seconds', '0', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['rccl_ranks'] == 8 and rec['collective_backend'] == 'gloo' and rec['scaling'] == 'weak'
    assert rec['parity']['mismatches'] == 0 and rec['parity']['ranks_checked'] == 8 and rec['parity']['ranks_failed'] == []
    assert rec['parity']['mismatches_all_ranks'] == 0
    # what the first real multi-GPU run must show (round-5 review item 4): every rank's own clock and kernel time, the slowest one priced
    assert len(rec['per_rank_ms_per_step']) == 8 and max(rec['per_rank_ms_per_step']) <= rec['ms_per_step'] * 1.0001
    assert len(rec['roofline']['kernel_ms_per_rank']) == 8 and abs(rec['roofline']['kernel_ms'] - max(rec['roofline']['kernel_ms_per_rank'])) < 1e-4
    direct = rec['cvq'].pop('direct_route')
    cluster = rec['cvq'].pop('cluster_sync_6272')
    for toks, blk in rec['cvq'].items():
        assert blk['codebook_in_sync'] is True and blk['collectives_per_step'] == 1.0, (toks, blk)
        assert blk['one_call_forward'] is True
        assert blk['exchange_bytes_per_step'] >= 4 * (2 * 16384 + 4)
    # the cluster config's anchor: the key exchange — two collectives per step, 8 M + 4 (2K + 4 + M D) bytes, nothing like the gathered flow's
    m = cluster['exchange_rows']
    assert cluster['codebook_in_sync'] is True and cluster['one_call_forward'] is True and cluster['collectives_per_step'] in (1.0, 2.0)
    assert abs(cluster['exchange_bytes_per_step'] - (8 * m + 4 * (2 * 8192 + 4 + m * 768))) < 1e-6 or cluster['collectives_per_step'] < 2.0
    assert cluster['exchange_bytes_per_step'] < cluster['dense_exchange_bytes_per_step'] / 100
    # the second sub-block: the same step in a child process per rank under a timeout (on RCCL: VQHIP_ALLREDUCE=direct; the shared-GPU
    # rehearsal runs gloo, where only the plumbing — rendezvous of the children, their JSON, the agreement — can be exercised)
    assert direct['ok'] is True and direct['route_requested'] == 'torch' and direct['rccl_ranks'] == 8, direct
    assert direct['codebook_in_sync'] is True and direct['collectives_per_step'] == 1.0 and direct['ms_per_step'] > 0, direct


def test_bench_direct_route_timeout_is_a_reported_failure(tmp_path):
    """A child that does not finish in time (a rank hanging inside ncclCommInitRank on the first real multi-GPU run) is killed and
    becomes `cvq.direct_route.ok == false` with the reason — the line itself is printed, complete, with exit code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQ_BENCH_SHARE_GPU='1', VQ_BENCH_DIRECT_TIMEOUT='0.5', VQ_BENCH_CVQ_SETTLE='30')
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--images', '16',
                          '--min-seconds', '0', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    direct = rec['cvq']['direct_route']
    assert direct['ok'] is False and 'did not finish within' in direct['error'], direct
    assert rec['n_gpus'] == 2 and rec['parity']['ranks_checked'] == 2 and len(rec['per_rank_ms_per_step']) == 2
    assert all(blk['codebook_in_sync'] for k, blk in rec['cvq'].items() if k != 'direct_route')
    assert rec['cvq']['cluster_sync_6272']['one_call_forward'] is True


def test_callbacks_at_world_size_two(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))) for r in range(2))
    np.testing.assert_array_equal(r0['sparse_rows'], r1['sparse_rows'])                  # every rank sized the same exchange
    for k in r0:
        if k.endswith('w_new') or k.endswith('_p') or k in ('lazy_w', 'sparse_w') or k.startswith('sync_w'):
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


# ---- the module under DistributedDataParallel (configs/strategies/ddp.py:5-6; every shipped training config wraps its model) ----

def _ddp_worker(rank, world, port, outdir):
    import sys

    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import synth
    from toy_model import build_toy, train_steps
    from vector_quantization_amd.utils import is_sync

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DRY_RUN='1')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    K, D, C, B, HW, steps = 512, 32, 8, 4, 8, 2      # (two steps: rccl_ws1_child.py explains why not more)
    gen = synth.rng(77)
    w0 = torch.from_numpy(synth.unit_rows(gen.standard_normal((K, D), dtype=np.float32)))
    images_all = [torch.from_numpy(gen.standard_normal((world * B, C, HW, HW), dtype=np.float32)).to(dev) for _ in range(steps)]
    mine = [im[rank::world].contiguous() for im in images_all]
    rec = {}

    def manual_steps(model, images, autocast):
        """The bare module with the gradient averaging DDP does, by hand."""
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
        out_rec = []
        for image in images:
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
                out, q_loss, quant = model(image)
                loss = out.float().pow(2).mean() + q_loss
            loss.backward()
            for p in model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad /= world
            opt.step()
            out_rec.append((loss.detach().clone(), quant.detach().clone()))
        return out_rec

    for kind in ('vqgan', 'cvq'):
        for autocast in (False, True):
            tag = f'{kind}_{"bf16" if autocast else "fp32"}'
            model = build_toy(kind, K, D, w0, dev)
            ddp = DDP(model, device_ids=[0], find_unused_parameters=True)
            grads = []
            r_ddp = train_steps(ddp, mine, autocast=autocast, grads_out=grads)
            for n, p in model.named_parameters():
                assert is_sync(p.detach()), f'{tag}: {n} differs across the ranks'
            if kind == 'cvq':
                assert is_sync(model._quantizer.get_buffer('_probability'))
            bare = build_toy(kind, K, D, w0, dev)
            r_bare = manual_steps(bare, mine, autocast)
            # DDP wrapping changes nothing in the step (to the last bits only up to the arrival order of the float atomics in
            # the convolutions' and the codebook's weight gradients)
            for (la, qa), (lb, qb) in zip(r_ddp, r_bare):
                assert torch.equal(qa, qb) and abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb))), tag
            for (na, pa), (nb, pb) in zip(model.named_parameters(), bare.named_parameters()):
                assert na == nb
                torch.testing.assert_close(pa.detach(), pb.detach(), rtol=1e-5, atol=1e-6, msg=f'{tag}: {na}')
            # state-dict round trip through the wrapper's module
            sd = {k: v.clone() for k, v in ddp.module.state_dict().items()}
            again = build_toy(kind, K, D, torch.zeros_like(w0), dev)
            again.load_state_dict(sd)
            again.eval(); model.eval()
            with torch.no_grad():
                oa, _, qa = again(mine[0])
                ob, _, qb = model(mine[0])
            assert torch.equal(qa, qb) and torch.equal(oa, ob), f'{tag}: state-dict round trip'
            rec[f'{tag}_w'] = model._quantizer.embedding.weight.detach().cpu().numpy()
            if kind == 'vqgan' and not autocast:
                # the single-process run on the concatenated batch: same tokens, gradients equal within 1e-5 (DDP averages
                # the per-rank gradients of per-rank means; equal shards make that the global mean's gradient)
                single = build_toy(kind, K, D, w0, dev)
                g1 = []
                r_single = train_steps(single, images_all, grads_out=g1)
                for t in range(steps):
                    for n, gs in g1[t].items():
                        gd = grads[t]['module.' + n]
                        torch.testing.assert_close(gd, gs, rtol=1e-4, atol=1e-5, msg=f'{tag} step {t}: {n}')
                # tokens: the concatenated batch is ordered image-major, the shard rank::world picks whole images
                toks = r_single[0][1].reshape(world * B, -1)[rank::world].reshape(-1)
                assert torch.equal(toks, r_ddp[0][1].reshape(-1))
    np.savez(os.path.join(outdir, f'ddp_rank{rank}.npz'), **rec)
    dist.barrier()
    dist.destroy_process_group()


def test_quantizer_inside_ddp_at_world_size_two(tmp_path):
    """A toy VQ model (1x1 conv -> this quantizer -> 1x1 conv) wrapped in DistributedDataParallel(find_unused_parameters=True),
    world size 2 (gloo, shared cuda:0), with and without bf16 autocast, the VQGAN and the CVQ-VAE configs, two optimizer
    steps: parameters and codebook bit-identical across the ranks (DRY_RUN arms the reference's is_sync asserts inside the
    callbacks), the step identical to the bare module with hand-averaged gradients, gradients equal to the single-process
    run on the concatenated batch, state-dict round trip."""
    import torch.multiprocessing as mp
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(str(tmp_path), f'ddp_rank{r}.npz'))) for r in range(2))
    for k in r0:
        assert r0[k].tobytes() == r1[k].tobytes(), f'ranks disagree on {k}'
