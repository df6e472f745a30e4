"""Generate tests/golden/*.npz FROM THE REFERENCE'S OWN SOURCE FILES — run in the build container, commit the outputs.

    python -m oracle.make_golden            # all fixtures (≈1 min on 8 cores)

Every expected value below is produced by executing the reference's modules where they lie under /root/reference
(``oracle/ref_import.py``: the real ``VQGANQuantizer`` / ``VQKDQuantizer`` built from the reference's config dicts
through its own registries, with ``L2Distance``/``CosineDistance``, ``VQGANLoss``/``CommitmentLoss``,
``NormalizeCallback``, ``VQKDCallback``, ``CVQVAECallback``, ``NearestAnchor``/``MultinomialAnchor``/``CachedAnchor``,
``QuantStatistics``, ``ste``).  Each ``.npz`` carries ``spec['source'] == 'reference-import'`` and the reference
file:line ranges that computed it.  The script also asserts, case by case, that the restatement
``oracle/torch_ref.py`` (which travels to the GPU box as bench.py's CPU baseline and as the float reference of the GPU
tests) gives byte-identical results; ``tests/test_reference_pin.py`` repeats that check in the CPU suite.

What is NOT the reference's: ``todd.utils.ema/EMA`` and ``todd.models.losses.MSELoss(norm=)`` are un-vendored and
follow SURVEY.md §8c (``ref_import._ToddArithmetic``).

Inputs are regenerated from seeds by ``oracle/synth.py`` (their sha256 is stored); small known-answer cases store
their inputs verbatim.  Fixtures hold data only.
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_import, synth, torch_ref as tr

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
EMB = 'torch_nn_modules_sparse_Embedding'          # configs/vq/interface.py:7

CITE_ENCODE = ['vq/tasks/image_tokenization/models/quantizers/base.py:123-182', 'vq/algorithms/vq/quantizers.py:92-117',
               'vq/algorithms/vq/distances.py:28-46', 'vq/algorithms/vq/losses.py:41-127',
               'vq/tasks/image_tokenization/models/quantizers/utils/ste.py:9-10', 'vq/algorithms/vq/utils.py:13-52',
               'vq/algorithms/vq/callbacks/normalize.py:22-29', 'vq/algorithms/vq/callbacks/update.py:53-56']
CITE_VQKD = ['vq/algorithms/vqkd/quantizers/callbacks.py:26-129', 'vq/algorithms/vq/callbacks/normalize.py:22-29']
CITE_CVQ = ['vq/algorithms/cvqvae/quantizer_callback.py:60-105', 'vq/algorithms/cvqvae/anchors.py:41-166',
            'vq/algorithms/vq/utils.py:13-52']

# name, kind, seed, N, K, D, distance, normalize(NormalizeCallback), loss
ENCODE_CASES = [
    ('l2_c1_normal_s0',        'normal',       0,    1024, 1024,  256, 'L2',     False, 'vqgan'),
    ('l2_c1_normal_s3407',     'normal',       3407, 1024, 1024,  256, 'L2',     False, 'vqgan'),
    ('l2_c1_planted_s3407',    'planted',      3407, 1024, 1024,  256, 'L2',     False, 'vqgan'),
    ('l2_c1_vqganinit_s3407',  'vqgan_init',   3407, 1024, 1024,  256, 'L2',     False, 'vqgan'),
    ('l2_c2_bf16x_s3407',      'normal_bf16x', 3407, 512,  16384, 256, 'L2',     False, 'vqgan'),
    ('l2_c2_vqganinit_s3407',  'vqgan_init',   3407, 512,  16384, 256, 'L2',     False, 'vqgan'),
    ('l2_int_small',           'int',          1,    64,   48,    16,  'L2',     False, 'vqgan'),
    ('l2_int_c1',              'int',          2,    128,  1024,  256, 'L2',     False, 'vqgan'),
    ('l2_tiny_nonmm',          'normal',       5,    20,   20,    16,  'L2',     False, 'vqgan'),
    ('l2_n1_d8',               'normal',       6,    1,    64,    8,   'L2',     False, 'vqgan'),
    ('l2_d32',                 'normal',       7,    300,  777,   32,  'L2',     False, 'vqgan'),
    ('cos_c3_unit_s3407',      'unit',         3407, 3136, 8192,  32,  'Cosine', False, 'commitment_norm'),
    ('cos_c1_normal_s3407',    'normal',       3407, 1024, 1024,  256, 'Cosine', False, 'commitment_norm'),
    ('cos_int_small',          'int',          3,    64,   48,    16,  'Cosine', False, 'commitment_norm'),
    ('norml2_llamagen_d8',     'normal',       3407, 2048, 16384, 8,   'L2',     True,  'vqgan'),
    ('normcos_vqkd_d32',       'normal',       11,   1024, 2048,  32,  'Cosine', True,  'commitment_norm'),
    # configs/cluster (CLIP/DINO/MAE/ViT features): D=768, K=8192, cosine, two 14x14 token maps
    ('cos_cluster_d768',       'normal',       768,  392,  8192,  768, 'Cosine', False, 'commitment_norm'),
    # the remaining proposal-kernel instantiations (D=512: 32 k-steps, D=1024: 64) and a ragged codebook
    ('l2_d512_ragged',         'normal',       512,  333,  1001,  512, 'L2',     False, 'vqgan'),
    ('l2_d1024',               'planted',      1024, 257,  700,   1024, 'L2',    False, 'vqgan'),
    # BASELINE configs[3] at full codebook size: CVQ-VAE default distance (cosine), per-rank batch 12x256 tokens
    ('cos_c4_k16384_s3407',    'normal',       3408, 3072, 16384, 256, 'Cosine', False, 'vqgan'),
]


# ---- the reference's quantizers, built from the reference's config dicts ----------------------------------------

def quantizer_config(K, D, distance, loss, callbacks=()):
    """configs/vqgan/model.py:19-23, configs/vqkd/model.py:20-26, configs/vq/{interface,distance,num_embeddings}.py."""
    cfg = dict(embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D), distance=dict(type=f'{distance}Distance'),
               callbacks=[dict(c) for c in callbacks])
    if loss == 'vqgan':
        cfg.update(type='VQGANQuantizer', losses=dict(vqgan_loss=dict(type='VQGANLoss')), init_weights=dict(type='vqgan'))
    elif loss == 'commitment_norm':
        cfg.update(type='VQKDQuantizer', losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))
    else:
        raise ValueError(loss)
    return cfg


def ref_quantizer(K, D, distance, loss, callbacks=(), w=None, train=False):
    ref = ref_import.load()
    q = ref.build_quantizer(quantizer_config(K, D, distance, loss, callbacks))
    q.eval()
    if any(c.get('type') == 'VQKDCallback' for c in callbacks):
        # the one-shot lazy-init hook (lazy_init_weights.py:28-37) fires on the first forward; in eval mode it returns
        # early (vqkd/quantizers/callbacks.py:84-85) and removes itself: the fixtures start from a GIVEN codebook
        q(torch.zeros(2, D), {})
        assert len(q._forward_pre_hooks) == 0
    if w is not None:
        q.embedding.weight.data = torch.as_tensor(w).clone()
    q.train(train)
    return q


def ref_forward(x, w, distance, loss, normalize):
    """One eval-mode BaseQuantizer.forward of the real reference module; returns the same dict as torch_ref.forward
    plus the distance matrix and the histogram."""
    K, D = w.shape
    q = ref_quantizer(K, D, distance, loss, [dict(type='NormalizeCallback')] if normalize else (), w)
    with torch.no_grad():
        z_ste, l, memo = q(torch.as_tensor(x), {})
        z, _ = q.decode(memo['quant'], {})
    ref = ref_import.load()
    hist = ref.QuantStatistics(quant=memo['quant'], codebook_size=K).bin_count()
    return dict(x=memo['x'], w=q.embedding.weight.detach(), quant=memo['quant'], z=z, z_ste=z_ste, loss=l,
                d=memo['encode']['distance'], hist=hist, loss_memo=dict(memo['loss']))


def same(a, b) -> bool:
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.numpy().tobytes() == b.numpy().tobytes()


def check_restatement(tag, got: dict, want: dict, keys):
    for k in keys:
        assert same(got[k], want[k]), f'{tag}: oracle/torch_ref.py differs from the reference import on {k!r}'


def encode_case(name, kind, seed, N, K, D, distance, normalize, loss):
    x, w = synth.make_inputs(kind, seed, N, K, D)
    out = ref_forward(x, w, distance, loss, normalize)
    rest = tr.forward(torch.from_numpy(x), torch.from_numpy(w), distance, loss, normalize=normalize)
    check_restatement(name, rest, out, ('x', 'w', 'quant', 'z', 'z_ste', 'loss'))
    assert same(tr.bin_count(rest['quant'], K), out['hist'])
    quant = out['quant'].numpy()
    mind = out['d'].gather(1, out['quant'].reshape(-1, 1)).reshape(-1).numpy()
    rec = dict(
        spec=json.dumps(dict(name=name, kind=kind, seed=seed, N=N, K=K, D=D, distance=distance,
                             normalize=normalize, loss=loss, torch=torch.__version__,
                             source='reference-import', reference=CITE_ENCODE)),
        x_sha=synth.sha(x), w_sha=synth.sha(w),
        quant=quant.astype(np.int32), mind=mind.astype(np.float32),
        loss=np.float32(out['loss'].item()),
        hist=out['hist'].numpy().astype(np.int32),
        z_sha=synth.sha(out['z'].numpy()), zste_sha=synth.sha(out['z_ste'].numpy()),
        z_head=out['z'].numpy()[:8], zste_head=out['z_ste'].numpy()[:8],
        col_idx=out['d'].argmin(0).numpy().astype(np.int32),          # NearestAnchor._anchors (cvqvae/anchors.py:83)
    )
    if N * D + K * D <= 1 << 14:   # small KATs carry their inputs
        rec['x'] = x
        rec['w'] = w
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    return rec


def autocast_cases():
    """CosineDistance.forward inside torch.autocast(bf16) — what the reference's trainers and validators run on a GPU
    (vq/runners/base.py:30-48 injects the autocast callback; F.normalize is on autocast's fp32 list, the einsum on its
    bf16 list): bf16 distance matrix, argmin with the lowest index on ties.  Executed here with the reference's own
    CosineDistance under CPU autocast; pins oracle `cos_bf16_argmin` (the opt-in VQHIP_METRIC_COS_BF16)."""
    ref = ref_import.load()
    dist = ref.CosineDistance()
    for name, kind, seed, N, K, D in (('cosbf16_c3_unit', 'unit', 3407, 1568, 8192, 32),
                                      ('cosbf16_normal_d256', 'normal', 12, 768, 4096, 256),
                                      ('cosbf16_int_small', 'int', 3, 64, 48, 16)):
        x, w = synth.make_inputs(kind, seed, N, K, D)
        with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
            d = dist(torch.from_numpy(x), torch.from_numpy(w))
        assert d.dtype == torch.bfloat16
        quant = d.argmin(-1)
        rec = dict(spec=json.dumps(dict(name=name, kind=kind, seed=seed, N=N, K=K, D=D, distance='CosineBF16',
                                        torch=torch.__version__, source='reference-import',
                                        reference=['vq/algorithms/vq/distances.py:35-46', 'vq/runners/base.py:30-48',
                                                   'vq/algorithms/vq/quantizers.py:97-99'])),
                   x_sha=synth.sha(x), w_sha=synth.sha(w), quant=quant.numpy().astype(np.int32),
                   mind=d.gather(1, quant.reshape(-1, 1)).reshape(-1).float().numpy(),
                   col_idx=d.argmin(0).numpy().astype(np.int32),
                   differs_from_fp32=np.int32(int((quant != dist(torch.from_numpy(x), torch.from_numpy(w)).argmin(-1)).sum())))
        if N * D + K * D <= 1 << 14:
            rec['x'] = x; rec['w'] = w
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
        print(f"{name:28s} rows that differ from the fp32 argmin: {int(rec['differs_from_fp32'])}/{N}", flush=True)


def autocast_module_cases():
    """The UNCHANGED VQ-KD and CVQ-VAE quantizer configs (configs/vqkd/model.py:20-26 with configs/vq/interface.py's
    `distance='Cosine'`; configs/cvqvae/quantizer.py:1-6) executed the way the reference's GPU trainers execute them: every
    step inside torch.autocast(bf16) (vq/runners/base.py:30-48 appends the AutocastCallback).  Real reference modules,
    CPU autocast on fp32 inputs — on this path CPU and CUDA autocast cast the same ops (einsum -> bf16; cdist / mse_loss
    -> fp32; F.normalize stays fp32 because its input is fp32).  Pins the product's `CosineDistance(autocast='auto')`."""
    x, w = update_inputs()
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    ac = lambda: torch.autocast('cpu', dtype=torch.bfloat16)    # noqa: E731
    # eval-mode forward of the VQ-KD quantizer (a validator's step)
    q = ref_quantizer(UPD['K'], UPD['D'], 'Cosine', 'commitment_norm', VQKD_CB, w)
    with torch.no_grad(), ac():
        z_eval, loss_eval, memo = q(xt, {})
    assert memo['encode']['distance'].dtype == torch.bfloat16
    quant_eval = memo['quant']
    with torch.no_grad():
        quant_fp32 = q(xt, {})[2]['quant']
    # train steps
    with ac():
        o = vqkd_step(x, w)
        r = tr.vqkd_train_step(xt, wt, UPD['ema_decay'])
    check_restatement('autocast_vqkd', r, o, ('quant', 'w_new', 'loss', 'grad_x'))
    with ac():
        s1, s2 = cvq_steps(x, w, 'Cosine')
        rc = tr.cvq_train_steps(xt, wt, 'Cosine', UPD['ema_decay'], UPD['eps'], steps=2)
    for got, want, tag in ((rc[0], s1, 'step1'), (rc[1], s2, 'step2')):
        check_restatement(f'autocast_cvq.{tag}', got, want, ('quant', 'col_idx', 'p', 'w_new', 'anchors', 'decay'))
    np.savez_compressed(
        os.path.join(OUT, 'autocast_modules.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w),
        eval_quant=quant_eval.numpy().astype(np.int32), eval_loss=np.float32(loss_eval.item()),
        eval_differs_from_fp32=np.int32(int((quant_eval != quant_fp32).sum())),
        vqkd_quant=o['quant'].numpy().astype(np.int32), vqkd_w_new=o['w_new'].numpy(), vqkd_loss=np.float32(o['loss'].item()),
        vqkd_grad_x_head=o['grad_x'].numpy()[:8], vqkd_grad_x_sha=synth.sha(o['grad_x'].numpy()),
        cvq_quant=s1['quant'].numpy().astype(np.int32), cvq_col_idx=s1['col_idx'].numpy().astype(np.int32),
        cvq_p1=s1['p'].numpy(), cvq_w_new=s1['w_new'].numpy(),
        cvq_quant2=s2['quant'].numpy().astype(np.int32), cvq_col_idx2=s2['col_idx'].numpy().astype(np.int32),
        cvq_p2=s2['p'].numpy(), cvq_w_new2=s2['w_new'].numpy(),
        spec=json.dumps(dict(UPD, distance='Cosine', autocast='bf16', source='reference-import',
                             reference=['vq/runners/base.py:30-48', 'vq/algorithms/vq/distances.py:35-46'] + CITE_VQKD + CITE_CVQ,
                             configs=['configs/vqkd/model.py:20-26', 'configs/cvqvae/quantizer.py:1-6'])))
    print(f"autocast_modules            eval rows that differ from the fp32 argmin: {int((quant_eval != quant_fp32).sum())}/{UPD['N']}",
          flush=True)


def special_case():
    """NaN / Inf handling of cdist + argmin (torch: NaN is the minimum, first NaN wins) through the real _encode."""
    N, K, D = 32, 64, 32
    x, w = synth.make_inputs('normal', 21, N, K, D)
    x[3, 5] = np.nan
    x[7, 0] = np.inf
    x[9, 2] = -np.inf
    w2 = w.copy()
    w2[10, 1] = np.nan                      # a NaN code poisons its whole column
    outs = []
    for ww, dist in ((w, 'L2'), (w2, 'L2'), (w, 'Cosine')):
        q = ref_quantizer(K, D, dist, 'vqgan', (), ww)
        with torch.no_grad():
            quant = q.encode(torch.from_numpy(x), {})[1]
        assert same(quant, tr.encode(torch.from_numpy(x), torch.from_numpy(ww), dist)[0])
        outs.append(quant.numpy().astype(np.int32))
    np.savez_compressed(os.path.join(OUT, 'special_nonfinite.npz'), x=x, w=w, w_nan=w2,
                        quant_l2=outs[0], quant_l2_wnan=outs[1], quant_cos=outs[2],
                        spec=json.dumps(dict(source='reference-import', reference=CITE_ENCODE[1:3])))


# ---- codebook updates: the real callbacks in train mode, one rank and two gloo ranks ----------------------------

UPD = dict(N=2048, K=512, D=32, seed=31, ema_decay=0.99, eps=1e-3)
VQKD_CB = [dict(type='VQKDCallback', ema=dict())]                                     # configs/vqkd/model.py:22


def cvq_cb(anchor='NearestAnchor', sync=False):
    return [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type=anchor, sync=sync))]  # configs/cvqvae/quantizer.py:1-5


def update_inputs():
    x, w = synth.make_inputs('normal', UPD['seed'], UPD['N'], UPD['K'], UPD['D'])
    return x, synth.unit_rows(w)


def trace_ema():
    """Record the (anchors, decay) the reference hands to todd.utils.ema (quantizer_callback.py:102)."""
    ref = ref_import.load()
    calls = []
    orig = ref.todd.utils.ema

    def spy(a, b, decay):
        calls.append((b.clone(), torch.as_tensor(decay).clone()))
        return orig(a, b, decay)

    ref.todd.utils.ema = spy
    return calls, lambda: setattr(ref.todd.utils, 'ema', orig)


def vqkd_step(x, w):
    q = ref_quantizer(UPD['K'], UPD['D'], 'Cosine', 'commitment_norm', VQKD_CB, w, train=True)
    xt = torch.as_tensor(x).clone().requires_grad_(True)
    z, loss, memo = q(xt, {})
    loss.backward()
    return dict(quant=memo['quant'], w_new=q.embedding.weight.detach().clone(), loss=loss.detach(), grad_x=xt.grad,
                z=z.detach())


def cvq_steps(x, w, dist, anchor='NearestAnchor', sync=False, steps=2):
    q = ref_quantizer(UPD['K'], UPD['D'], dist, 'vqgan', cvq_cb(anchor, sync), w, train=True)
    calls, undo = trace_ema()
    outs = []
    try:
        for _ in range(steps):
            with torch.no_grad():
                z, loss, memo = q(torch.as_tensor(x), {})
            anchors, decay = calls[-1]
            outs.append(dict(quant=memo['quant'], col_idx=memo['encode']['distance'].argmin(0),
                             p=q.get_buffer('_probability').clone(), w_new=q.embedding.weight.detach().clone(),
                             anchors=anchors, decay=decay))
    finally:
        undo()
    return outs


def _rank_worker(rank, world, port, outdir):
    """One gloo rank running the REAL reference callbacks on rows rank::world (DistributedSampler-style shards)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DRY_RUN='1')   # DRY_RUN: the reference's is_sync asserts run
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(0)                  # identical nn.Embedding init on every rank (the reference's DDP broadcast)
    x, w = update_inputs()
    xr = x[rank::world]
    rec = {}
    o = vqkd_step(xr, w)
    rec['vqkd_quant'], rec['vqkd_w_new'] = o['quant'].numpy(), o['w_new'].numpy()
    for dname in ('L2', 'Cosine'):
        for sync in (False, True):
            o = cvq_steps(xr, w, dname, sync=sync, steps=1)[0]
            tag = f'cvq_{dname.lower()}_{"sync" if sync else "avg"}'
            rec[f'{tag}_w_new'], rec[f'{tag}_p'] = o['w_new'].numpy(), o['p'].numpy()
            rec[f'{tag}_col_idx'] = o['col_idx'].numpy()
            rec[f'{tag}_quant'] = o['quant'].numpy()
    # lazy k-means init across ranks: gather to rank 0, Lloyd iterations there, broadcast (callbacks.py:77-112)
    # (DRY_RUN off here: inside the Lloyd loop only rank 0 calls _update_embedding, whose DRY_RUN is_sync assert is a
    #  collective — the reference itself would hang there in a multi-rank dry run)
    os.environ['DRY_RUN'] = ''
    random.seed(1234)
    ref = ref_import.load()
    q = ref.build_quantizer(quantizer_config(64, UPD['D'], 'Cosine', 'commitment_norm', VQKD_CB))
    q.train()
    q(torch.from_numpy(xr[:256]), {})
    rec['lazy_w'] = q.embedding.weight.detach().numpy()
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **rec)
    dist.barrier()
    dist.destroy_process_group()


def two_rank_reference():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_worker, args=(2, 29731, d), nprocs=2, join=True)
        return [dict(np.load(os.path.join(d, f'rank{r}.npz'))) for r in range(2)]


def eight_rank_reference():
    """The reference's callbacks at the world size its shipped configs train at (8 ranks: configs/strategies, batch_size_in_total)."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_worker, args=(8, 29741, d), nprocs=8, join=True)
        return [dict(np.load(os.path.join(d, f'rank{r}.npz'))) for r in range(8)]


def update_cases():
    x, w = update_inputs()
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    ranks = two_rank_reference()
    for k in ranks[0]:
        if k.endswith('w_new') or k.endswith('_p') or k == 'lazy_w':
            assert ranks[0][k].tobytes() == ranks[1][k].tobytes(), f'reference ranks disagree on {k}'
    # --- VQ-KD: NormalizeCallback.before_encode + VQKDCallback.after_encode ---
    o = vqkd_step(x, w)
    r = tr.vqkd_train_step(xt, wt, UPD['ema_decay'])
    check_restatement('update_vqkd', r, o, ('quant', 'w_new', 'loss', 'grad_x'))
    r2 = tr.vqkd_train_step_2rank(xt, wt, UPD['ema_decay'])
    assert same(r2['w_new'], ranks[0]['vqkd_w_new']), 'torch_ref 2-rank VQ-KD emulation differs from the 2-process reference run'
    np.savez_compressed(
        os.path.join(OUT, 'update_vqkd.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w),
        quant=o['quant'].numpy().astype(np.int32), w_new=o['w_new'].numpy(), loss=np.float32(o['loss'].item()),
        grad_x_sha=synth.sha(o['grad_x'].numpy()), grad_x_head=o['grad_x'].numpy()[:8],
        w_new_2rank=ranks[0]['vqkd_w_new'], quant_rank0=ranks[0]['vqkd_quant'].astype(np.int32),
        quant_rank1=ranks[1]['vqkd_quant'].astype(np.int32),
        spec=json.dumps(dict(UPD, source='reference-import', reference=CITE_VQKD,
                             two_rank='two gloo processes running the reference callbacks on rows r::2')))
    # --- CVQ-VAE (L2 and cosine), NearestAnchor ---
    for dist in ('L2', 'Cosine'):
        s1, s2 = cvq_steps(x, w, dist)
        r = tr.cvq_train_steps(xt, wt, dist, UPD['ema_decay'], UPD['eps'], steps=2)
        for got, want, tag in ((r[0], s1, 'step1'), (r[1], s2, 'step2')):
            check_restatement(f'update_cvq_{dist}.{tag}', got, want, ('quant', 'col_idx', 'p', 'w_new', 'anchors', 'decay'))
        tag = f'cvq_{dist.lower()}'
        r2 = tr.cvq_train_step_2rank(xt, wt, dist, UPD['ema_decay'], UPD['eps'])
        assert same(r2['avg']['w_new'], ranks[0][f'{tag}_avg_w_new']) and same(r2['avg']['p'], ranks[0][f'{tag}_avg_p'])
        assert same(r2['sync']['w_new'], ranks[0][f'{tag}_sync_w_new'])
        np.savez_compressed(
            os.path.join(OUT, f'update_cvq_{dist.lower()}.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w),
            quant=s1['quant'].numpy().astype(np.int32), col_idx=s1['col_idx'].numpy().astype(np.int32),
            p1=s1['p'].numpy(), decay=s1['decay'].numpy(), w_new=s1['w_new'].numpy(),
            quant2=s2['quant'].numpy().astype(np.int32), col_idx2=s2['col_idx'].numpy().astype(np.int32),
            p2=s2['p'].numpy(), w_new2=s2['w_new'].numpy(),
            w_new_2rank=ranks[0][f'{tag}_avg_w_new'], p_2rank=ranks[0][f'{tag}_avg_p'],
            col_idx_rank0=ranks[0][f'{tag}_avg_col_idx'].astype(np.int32),
            col_idx_rank1=ranks[1][f'{tag}_avg_col_idx'].astype(np.int32),
            w_new_2rank_sync=ranks[0][f'{tag}_sync_w_new'], p_2rank_sync=ranks[0][f'{tag}_sync_p'],
            col_idx_2rank_sync=ranks[0][f'{tag}_sync_col_idx'].astype(np.int32),
            spec=json.dumps(dict(UPD, distance=dist, source='reference-import', reference=CITE_CVQ,
                                 two_rank='two gloo processes running the reference callbacks on rows r::2; '
                                          '"avg" = NearestAnchor(sync=False) (anchors.py:64-67), '
                                          '"sync" = NearestAnchor(sync=True) (anchors.py:50-57)')))
    # --- the same at eight ranks (rows r::8): codebooks, probabilities, per-rank tokens and column argmins ---
    r8 = eight_rank_reference()
    for k in r8[0]:
        if k.endswith('w_new') or k.endswith('_p') or k == 'lazy_w':
            assert all(r8[0][k].tobytes() == r[k].tobytes() for r in r8[1:]), f'reference ranks disagree on {k} at world size 8'
    rec8 = {k: r8[0][k] for k in r8[0] if k.endswith('w_new') or k.endswith('_p') or k == 'lazy_w'}
    rec8['vqkd_quant'] = np.stack([r['vqkd_quant'] for r in r8]).astype(np.int32)
    for tag in ('cvq_l2_avg', 'cvq_cosine_avg', 'cvq_l2_sync', 'cvq_cosine_sync'):
        rec8[f'{tag}_quant'] = np.stack([r[f'{tag}_quant'] for r in r8]).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, 'update_8rank.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w), **rec8,
                        spec=json.dumps(dict(UPD, world=8, source='reference-import', reference=CITE_VQKD + CITE_CVQ,
                                             eight_rank='eight gloo processes running the reference callbacks on rows r::8 '
                                                        '(VQ-KD step; CVQ-VAE step with NearestAnchor sync False / True, L2 and cosine; '
                                                        'k-means lazy init from 256 rows per rank, K = 64)')))
    np.savez_compressed(os.path.join(OUT, 'lazy_init_2rank.npz'), w=ranks[0]['lazy_w'],
                        spec=json.dumps(dict(K=64, D=UPD['D'], N_per_rank=256, seed=1234, source='reference-import',
                                             reference=['vq/algorithms/vqkd/quantizers/callbacks.py:26-35,77-112'])))


# ---- k-means lazy init (single rank), alternative anchors, EntropyLoss -------------------------------------------

def lazy_init_case():
    """VQKDCallback.lazy_init_weights (callbacks.py:77-112) with a seeded random.sample; per-iteration assignments are
    recorded by wrapping the reference quantizer's _encode."""
    N, K, D, seed = 4096, 256, 32, 1234
    x, _ = synth.make_inputs('normal', 41, N, K, D)
    ref = ref_import.load()
    q = ref.build_quantizer(quantizer_config(K, D, 'Cosine', 'commitment_norm', VQKD_CB))
    w0 = q.embedding.weight.detach().clone()
    q.train()
    quants, books = [], []
    inner = q._encode

    def spy(xx, memo):
        quant, memo = inner(xx, memo)
        quants.append(quant.clone())
        books.append(q.embedding.weight.detach().clone())
        return quant, memo

    q._encode = spy
    after_init = []
    # registered after the one-shot lazy-init hook, so it sees the codebook exactly as lazy_init_weights left it
    q.register_forward_pre_hook(lambda m, a: after_init.append(m.embedding.weight.detach().clone()))
    random.seed(seed)
    with torch.no_grad():
        q(torch.from_numpy(x), {})
    assert len(quants) == 11                                    # 10 Lloyd iterations + the forward's own encode
    random.seed(seed)
    r = tr.vqkd_lazy_init(torch.from_numpy(x), w0, iters=10)
    assert same(r['indices'], torch.as_tensor(random.Random(seed).sample(range(N), K)))
    for i in range(10):
        assert same(r['quants'][i], quants[i]), f'lazy init: torch_ref differs from the reference at Lloyd iteration {i}'
    assert same(r['w'], after_init[0])
    np.savez_compressed(
        os.path.join(OUT, 'lazy_init_vqkd.npz'), x_sha=synth.sha(x), indices=r['indices'].numpy().astype(np.int32),
        quants=torch.stack(quants[:10]).numpy().astype(np.int16), books_first=books[0].numpy(), w_init=after_init[0].numpy(),
        spec=json.dumps(dict(N=N, K=K, D=D, x_seed=41, seed=seed, iters=10, source='reference-import',
                             reference=['vq/algorithms/vqkd/quantizers/callbacks.py:77-112'])))


def anchor_cases():
    """MultinomialAnchor (anchors.py:88-104) and CachedAnchor (:107-166) called directly, as CVQVAECallback does."""
    ref = ref_import.load()
    N, K, D = 256, 64, 32
    x, w = synth.make_inputs('normal', 51, N, K, D)
    xt, wt = torch.from_numpy(x), torch.from_numpy(synth.unit_rows(w))
    d = ref.L2Distance()(xt, wt)
    quant = d.argmin(-1)
    p = torch.zeros(K)
    rec = dict(x_sha=synth.sha(x), w_sha=synth.sha(wt.numpy()))
    # Multinomial: the sampling distribution is the parity target (device RNG streams differ by design)
    torch.manual_seed(7)
    a, _ = ref.MultinomialAnchor()(xt, wt, d, quant, p)
    torch.manual_seed(7)
    idx = d.T.softmax(1).multinomial(1).reshape(-1)
    assert same(a, xt[idx])
    rec['multinomial_probs'] = d.T.softmax(1).numpy()
    rec['multinomial_idx_cpu_seed7'] = idx.numpy().astype(np.int32)
    # Cached: host-side RNG (random.sample / CPU randperm) => reproducible on any device
    ca = ref.CachedAnchor()
    random.seed(11)
    a1, _ = ca(xt, wt, d, quant, p)                                  # N > K: random.sample(range(N), K)
    rec['cached_big'] = a1.numpy()
    assert same(ca.cache, a1)
    torch.manual_seed(12)
    a2, _ = ca(xt[:K], wt, d[:K], quant[:K], p)                      # N == K: torch.randperm(K) on the CPU generator
    rec['cached_eq'] = a2.numpy()
    torch.manual_seed(13)
    a3, _ = ca(xt[:40], wt, d[:40], quant[:40], p)                   # N < K with a cache: cat([x, cache]) then random.sample
    rec['cached_small_with_cache'] = a3.numpy()
    rec['cached_small_seed'] = np.int32(13)
    np.savez_compressed(os.path.join(OUT, 'anchors_alt.npz'), **rec,
                        spec=json.dumps(dict(N=N, K=K, D=D, seed=51, python_seed_big=11, torch_seed_eq=12,
                                             python_seed_small=None, source='reference-import',
                                             reference=['vq/algorithms/cvqvae/anchors.py:88-166'])))


def entropy_case():
    """EntropyLoss (losses.py:130-153) over memo['distance'] with autograd through L2Distance / CosineDistance."""
    ref = ref_import.load()
    N, K, D = 128, 64, 16
    x, w = synth.make_inputs('normal', 3, N, K, D)
    rec = dict(x=x, w=w)
    for dist in ('L2', 'Cosine'):
        xt = torch.from_numpy(x).requires_grad_(True)
        wt = torch.from_numpy(w).requires_grad_(True)
        d = getattr(ref, f'{dist}Distance')()(xt, wt.clone())
        # (not in losses.py's __all__; reachable the way a config reaches it: through the loss registry)
        entropy = ref.VQITQuantizerLossRegistry.build(dict(type='EntropyLoss', temperature=0.5))
        loss = entropy(None, xt, dict(distance=d))
        loss.backward()
        r = tr.entropy_loss(torch.from_numpy(x), torch.from_numpy(w), dist, 0.5)
        assert same(r['loss'], loss.detach()) and same(r['grad_x'], xt.grad) and same(r['grad_w'], wt.grad)
        rec[f'loss_{dist.lower()}'] = np.float32(loss.item())
        rec[f'grad_x_{dist.lower()}'] = xt.grad.numpy()
        rec[f'grad_w_{dist.lower()}'] = wt.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'entropy_loss.npz'), **rec,
                        spec=json.dumps(dict(N=N, K=K, D=D, seed=3, temperature=0.5, source='reference-import',
                                             reference=['vq/algorithms/vq/losses.py:130-153',
                                                        'vq/algorithms/vq/distances.py:28-46'])))


def connector_case():
    """f3: post_encode ConvConnector -> BaseModel.quantize (the two einops rearrangements around the quantizer call,
    models/base.py:116-128) -> pre_decode ConvConnector, and BaseModel.encode_to_quant (:135-146), executed from the
    reference's own files.  BaseModel itself is abstract and drags encoders in: its two methods are called on a
    minimal holder object (they touch self._quantizer and self.encode only)."""
    ref = ref_import.load()
    B, Cin, D, H, W, K = 2, 16, 32, 8, 8, 256
    gen = synth.rng(61)
    x_in = gen.standard_normal((B, Cin, H, W), dtype=np.float32)
    wcb = gen.standard_normal((K, D), dtype=np.float32)
    torch.manual_seed(61)
    post = ref.VQITConnectorRegistry.build(dict(type='ConvConnector', in_channels=Cin, out_channels=D))      # configs/vqgan/model.py:18
    pre = ref.VQITConnectorRegistry.build(dict(type='ConvConnector', in_channels=D, out_channels=Cin))       # :24
    q = ref_quantizer(K, D, 'L2', 'vqgan', (), wcb)

    class Holder:
        _quantizer = q

        def encode(self, image, memo):
            return image, memo

    with torch.no_grad():
        x_map, _ = post(torch.from_numpy(x_in), {})
        z_map, q_loss, memo = ref.BaseModel.quantize(Holder(), x_map, {})
        out, _ = pre(z_map, {})
        quant_map, memo2 = ref.BaseModel.encode_to_quant(Holder(), x_map, {})
    assert same(quant_map.reshape(-1), memo['quantizer']['quant']) and tuple(memo['quantizer']['x_shape']) == (B, D, H, W)
    # restatement: the same two rearrangements around torch_ref.forward
    r = tr.forward(x_map.permute(0, 2, 3, 1).reshape(-1, D), torch.from_numpy(wcb), 'L2', 'vqgan')
    assert same(r['quant'], memo['quantizer']['quant']) and same(r['loss'], q_loss)
    assert same(r['z_ste'].reshape(B, H, W, D).permute(0, 3, 1, 2).contiguous(), z_map)
    np.savez_compressed(
        os.path.join(OUT, 'connector_path.npz'), x_in=x_in, w_sha=synth.sha(wcb),
        post_weight=post._conv.weight.detach().numpy(), post_bias=post._conv.bias.detach().numpy(),
        pre_weight=pre._conv.weight.detach().numpy(), pre_bias=pre._conv.bias.detach().numpy(),
        x_map=x_map.numpy(), z_map=z_map.numpy(), loss=np.float32(q_loss.item()),
        quant=quant_map.numpy().astype(np.int32), out=out.numpy(),
        spec=json.dumps(dict(B=B, Cin=Cin, D=D, H=H, W=W, K=K, seed=61, source='reference-import',
                             reference=['vq/tasks/image_tokenization/models/connectors/conv.py:15-56',
                                        'vq/tasks/image_tokenization/models/connectors/base.py:12-39',
                                        'vq/tasks/image_tokenization/models/base.py:116-146'])))


def runner_cases():
    """f1/f2: the runner-level callers, executed from the reference's own files (ref_import.load_runners): the token file
    TokenizeCallback writes, the two .npy files of the LlamaGen TokenizeCallback, the values CodebookUsageMetric /
    CodebookPPLMetric return over three iterations, and Tokenizer._run_iter on a model whose encode_to_quant is the
    reference BaseModel's.  The files are stored byte for byte (uint8 arrays)."""
    import pathlib
    import types
    import torch.distributed as dist
    ref, rr = ref_import.load(), ref_import.load_runners()
    B, C, H, W, K = 3, 32, 4, 4, 64
    gen = synth.rng(71)
    quant = gen.integers(0, K, size=B * H * W, dtype=np.int64)
    category = np.array([1, 207, 999], dtype=np.int64)
    ids = ['n01440764_10026', 'n02099601_7', 'n15075141_999']
    quant10 = gen.integers(0, K, size=10 * H * W, dtype=np.int64)
    label = np.array([388], dtype=np.int64)
    metric_quants = [gen.integers(0, K // 2 + 8 * i, size=40 + 16 * i, dtype=np.int64) for i in range(3)]
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        runner = types.SimpleNamespace(work_dir=pathlib.Path(tmp), iter_=3, dataset=types.SimpleNamespace(image_size=256))
        cb = rr.TokenizeCallback()
        cb.bind(runner)
        cb.after_run_iter(dict(id_=ids, category=torch.from_numpy(category)),
                          dict(quantizer=dict(quant=torch.from_numpy(quant), x_shape=torch.Size((B, C, H, W)))))
        rec['tokens_pth'] = np.frombuffer((pathlib.Path(tmp) / 'tokens' / '3_0.pth').read_bytes(), dtype=np.uint8)
        cb2 = rr.LlamaGenTokenizeCallback()
        cb2.bind(runner)
        batch = dict(original_image=torch.zeros(1, 10, 3, 8, 8, dtype=torch.uint8), image=torch.zeros(1, 10, 3, 8, 8),
                     category=torch.from_numpy(label))
        cb2.before_run_iter(batch, {})
        assert batch['image'].shape == (10, 3, 8, 8)
        cb2.after_run_iter(batch, dict(quantizer=dict(quant=torch.from_numpy(quant10))))
        base = pathlib.Path(tmp) / 'llamagen_tokens'
        rec['llamagen_codes_npy'] = np.frombuffer((base / 'imagenet256_codes' / '2.npy').read_bytes(), dtype=np.uint8)
        rec['llamagen_labels_npy'] = np.frombuffer((base / 'imagenet256_labels' / '2.npy').read_bytes(), dtype=np.uint8)
        # metrics: summary() all-reduces, which needs a process group — one rank, gloo
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group('gloo', store=dist.FileStore(os.path.join(tmp, 'store'), 1), rank=0, world_size=1)
        try:
            runner.strategy = types.SimpleNamespace(module=types.SimpleNamespace(quantizer=types.SimpleNamespace(codebook_size=K)))
            usage = rr.CodebookUsageMetric(quant='["quantizer"]["quant"]')        # configs/vqgan/runner.py:121-128
            ppl = rr.CodebookPPLMetric(quant='["quantizer"]["quant"]')
            assert usage.summary({}) == 0. and ppl.summary({}) == 0.              # before any iteration (metrics.py:50-51)
            for m in (usage, ppl):
                m.bind(runner)
                for mq in metric_quants:
                    m.forward({}, dict(quantizer=dict(quant=torch.from_numpy(mq))))
            rec['usage'] = np.float64(usage.summary({}))
            rec['ppl'] = np.float64(ppl.summary({}))
            rec['counts'] = usage._counts.numpy().astype(np.int64)
        finally:
            if own_group:
                dist.destroy_process_group()
    # Tokenizer._run_iter (tokenizer.py:44-55) on a holder whose encode_to_quant is BaseModel's (models/base.py:130-146)
    Kq, Dq = 128, 16
    wcb = gen.standard_normal((Kq, Dq), dtype=np.float32)
    image = gen.standard_normal((2, Dq, 4, 4), dtype=np.float32)
    q = ref_quantizer(Kq, Dq, 'L2', 'vqgan', (), wcb)

    class Holder:
        _quantizer = q

        def encode(self, image, memo):
            return image, memo

        def encode_to_quant(self, image, memo):
            return ref.BaseModel.encode_to_quant(self, image, memo)

    tok = object.__new__(rr.Tokenizer)
    tok.strategy = types.SimpleNamespace(module=Holder())
    with torch.no_grad():
        memo = rr.Tokenizer._run_iter(tok, dict(original_image=torch.from_numpy(image), image=torch.from_numpy(image)), {})
    assert sorted(memo) == ['image', 'original_image', 'quantizer'] and tuple(memo['quantizer']['x_shape']) == (2, Dq, 4, 4)
    np.savez_compressed(
        os.path.join(OUT, 'runner_files.npz'), **rec, quant=quant, category=category, quant10=quant10, label=label,
        metric_quants_0=metric_quants[0], metric_quants_1=metric_quants[1], metric_quants_2=metric_quants[2],
        tok_image=image, tok_w=wcb, tok_quant=memo['quantizer']['quant'].numpy().astype(np.int64),
        spec=json.dumps(dict(B=B, C=C, H=H, W=W, K=K, seed=71, ids=ids, iter=3, rank=0, image_size=256, source='reference-import',
                             torch=torch.__version__,
                             reference=['vq/tasks/image_tokenization/runners/callbacks.py:23-53',
                                        'vq/tasks/image_tokenization/runners/metrics.py:25-73',
                                        'vq/tasks/image_tokenization/runners/tokenizer.py:44-55',
                                        'vq/tasks/image_tokenization/models/base.py:130-146',
                                        'tools/tokenize_llamagen.py:65-103'])))


def main(only=(), out_dir=None, quiet=False):
    global OUT
    if not ref_import.available():
        sys.exit('oracle/make_golden.py needs /root/reference (build container only)')
    if out_dir is not None:
        OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    only = set(only)
    print_ = (lambda *a, **k: None) if quiet else print
    for c in ENCODE_CASES:
        if only and c[0] not in only:
            continue
        rec = encode_case(*c)
        print_(f'{c[0]:28s} loss={float(rec["loss"]):.6f} used={int((rec["hist"] > 0).sum())}/{c[4]}', flush=True)
    if not only or 'autocast' in only:
        autocast_cases()
        autocast_module_cases()
    if not only or 'special' in only:
        special_case()
    if not only or 'update' in only:
        update_cases()
    if not only or 'lazy' in only:
        lazy_init_case()
    if not only or 'anchors' in only:
        anchor_cases()
    if not only or 'entropy' in only:
        entropy_case()
    if not only or 'connector' in only:
        connector_case()
    if not only or 'runners' in only:
        runner_cases()
    print_('fixtures written to', OUT, '— every value produced by the reference files:')
    for n, f in ref_import.load().files.items():
        print_('   ', f)


if __name__ == '__main__':
    main(sys.argv[1:])
