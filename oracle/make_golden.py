"""Generate tests/golden/*.npz — run in the BUILD CONTAINER (CPU torch), commit the outputs.

    python -m oracle.make_golden

The reference package itself is not importable here (needs todd_ai + python>=3.11); the
fixtures are produced by ``oracle/torch_ref.py``, which calls the same ATen ops in the same order as
the reference's quantizer.  Inputs are regenerated from seeds by ``oracle/synth.py`` (their sha256 is
stored); small known-answer cases store their inputs verbatim.  Fixtures hold data only.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import synth, torch_ref as tr

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

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
]


def encode_case(name, kind, seed, N, K, D, distance, normalize, loss):
    x, w = synth.make_inputs(kind, seed, N, K, D)
    out = tr.forward(torch.from_numpy(x), torch.from_numpy(w), distance, loss, normalize=normalize)
    quant = out['quant'].numpy()
    xe = out['x']
    d = tr.DISTANCES[distance](xe, out['w'])
    mind = d.gather(1, out['quant'].reshape(-1, 1)).reshape(-1).numpy()
    rec = dict(
        spec=json.dumps(dict(name=name, kind=kind, seed=seed, N=N, K=K, D=D, distance=distance,
                             normalize=normalize, loss=loss, torch=torch.__version__)),
        x_sha=synth.sha(x), w_sha=synth.sha(w),
        quant=quant.astype(np.int32), mind=mind.astype(np.float32),
        loss=np.float32(out['loss'].item()),
        hist=tr.bin_count(out['quant'], K).numpy().astype(np.int32),
        z_sha=synth.sha(out['z'].numpy()), zste_sha=synth.sha(out['z_ste'].numpy()),
        z_head=out['z'].numpy()[:8], zste_head=out['z_ste'].numpy()[:8],
    )
    if N * D + K * D <= 1 << 14:   # small KATs carry their inputs
        rec['x'] = x
        rec['w'] = w
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    return rec


def special_case():
    """NaN / Inf handling of cdist + argmin (torch: NaN is the minimum, first NaN wins)."""
    N, K, D = 32, 64, 32
    x, w = synth.make_inputs('normal', 21, N, K, D)
    x[3, 5] = np.nan
    x[7, 0] = np.inf
    x[9, 2] = -np.inf
    w2 = w.copy()
    w2[10, 1] = np.nan                      # a NaN code poisons its whole column
    q1 = tr.encode(torch.from_numpy(x), torch.from_numpy(w), 'L2')[0].numpy()
    q2 = tr.encode(torch.from_numpy(x), torch.from_numpy(w2), 'L2')[0].numpy()
    q3 = tr.encode(torch.from_numpy(x), torch.from_numpy(w), 'Cosine')[0].numpy()
    np.savez_compressed(os.path.join(OUT, 'special_nonfinite.npz'), x=x, w=w, w_nan=w2,
                        quant_l2=q1.astype(np.int32), quant_l2_wnan=q2.astype(np.int32),
                        quant_cos=q3.astype(np.int32))


def update_cases():
    """One VQ-KD k-means/EMA step and one CVQ-VAE step, single rank and emulated 2-rank sums (F4)."""
    N, K, D = 2048, 512, 32
    x, w = synth.make_inputs('normal', 31, N, K, D)
    w = synth.unit_rows(w)
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    # --- VQ-KD (cosine, NormalizeCallback semantics: x normalised before encode) ---
    xn = torch.nn.functional.normalize(xt)
    quant, _ = tr.encode(xn, wt, 'Cosine')
    w1 = tr.vqkd_after_encode(xn, quant, wt, ema_decay=0.99)
    # 2 ranks: rank r holds rows r::2 ; the all-reduced hist / sums are the sums of the per-rank ones
    hs, ss = [], []
    for r in range(2):
        xr, qr = xn[r::2], quant[r::2]
        hs.append(tr.bin_count(qr, K))
        c = torch.zeros_like(wt)
        c.scatter_add_(0, qr.reshape(-1, 1).expand(-1, D), torch.nn.functional.normalize(xr))
        ss.append(c)
    w1_2rank = tr.vqkd_after_encode(xn[0::2], quant[0::2], wt, 0.99, hs[0] + hs[1], ss[0] + ss[1])
    np.savez_compressed(os.path.join(OUT, 'update_vqkd.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w),
                        quant=quant.numpy().astype(np.int32), w_new=w1.numpy(),
                        w_new_2rank=w1_2rank.numpy(),
                        spec=json.dumps(dict(N=N, K=K, D=D, seed=31, ema_decay=0.99)))
    # --- CVQ-VAE (L2 and cosine variants), NearestAnchor, sync=False ---
    for dist in ('L2', 'Cosine'):
        quant, d = tr.encode(xt, wt, dist)
        p0 = torch.zeros(K)
        w_new, p1, anchors, indices, decay = tr.cvq_after_encode(xt, quant, d, wt, p0, 0.99, 1e-3)
        # second step from the updated state (p no longer zero)
        quant2, d2 = tr.encode(xt, w_new, dist)
        w_new2, p2, _, indices2, _ = tr.cvq_after_encode(xt, quant2, d2, w_new, p1, 0.99, 1e-3)
        # emulated 2 ranks (sync=False): stats all-reduced, anchors averaged
        halves = []
        for r in range(2):
            xr = xt[r::2]
            qr, dr = tr.encode(xr, wt, dist)
            halves.append((xr, qr, dr))
        hist = sum(tr.bin_count(h[1], K) for h in halves)
        numel = torch.tensor(N)
        a_other = tr.nearest_anchor(halves[1][0], halves[1][2])[0]
        w_2r, p_2r, _, idx_r0, _ = tr.cvq_after_encode(
            halves[0][0], halves[0][1], halves[0][2], wt, p0, 0.99, 1e-3,
            world_hist=hist, world_numel=numel, world_size=2, other_anchors=[a_other])
        np.savez_compressed(
            os.path.join(OUT, f'update_cvq_{dist.lower()}.npz'), x_sha=synth.sha(x), w_sha=synth.sha(w),
            quant=quant.numpy().astype(np.int32), col_idx=indices.numpy().astype(np.int32),
            p1=p1.numpy(), decay=decay.numpy(), w_new=w_new.numpy(),
            quant2=quant2.numpy().astype(np.int32), col_idx2=indices2.numpy().astype(np.int32),
            p2=p2.numpy(), w_new2=w_new2.numpy(),
            w_new_2rank=w_2r.numpy(), p_2rank=p_2r.numpy(), col_idx_rank0=idx_r0.numpy().astype(np.int32),
            spec=json.dumps(dict(N=N, K=K, D=D, seed=31, ema_decay=0.99, eps=1e-3, distance=dist)))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    for c in ENCODE_CASES:
        rec = encode_case(*c)
        print(f'{c[0]:28s} loss={float(rec["loss"]):.6f} used={int((rec["hist"] > 0).sum())}/{c[4]}')
    special_case()
    update_cases()
    print('fixtures written to', OUT)


if __name__ == '__main__':
    main()
