"""The reference's quantizer API re-hosted on libvqhip: configs build, hooks fire in the reference's order, and
forward / backward / codebook updates match the oracle (indices bit-exact, floats within 1e-5) and the golden
fixtures produced by the reference's ATen ops."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co, synth, torch_ref as tr

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

EMB = 'torch_nn_modules_sparse_Embedding'      # configs/vq/interface.py:7


def build(cfg, train=False, init=None):
    from vector_quantization_amd import Config, build_quantizer
    q = build_quantizer(cfg)
    q.train(train)
    q.init_weights(Config(init or {}))
    return q.cuda()


def vqgan_cfg(K, D, distance='L2', **extra):
    # configs/vqgan/model.py:19-23 + configs/vq/{interface,distance}.py
    return dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                distance=dict(type=f'{distance}Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')), **extra)


def set_weight(q, w):
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(w))


def ref_grads(x, w, quant, beta=0.25, upstream=None):
    """Gradients of the reference composition (decode → VQGANLoss → STE) on CPU with the given indices."""
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    z = tr.decode(torch.from_numpy(quant), wt)
    loss = tr.vqgan_loss(z, xt, beta)
    zs = tr.ste(z, xt)
    total = loss + (zs * upstream).sum() if upstream is not None else loss
    total.backward()
    return loss.item(), zs.detach().numpy(), xt.grad.numpy(), wt.grad.numpy()


@pytest.mark.parametrize('fused', [True, False])
def test_vqgan_forward_backward(fused):
    N, K, D = 1024, 1024, 256
    x, w = synth.make_inputs('normal', 3407, N, K, D)
    q = build(vqgan_cfg(K, D, fused=fused), train=True, init=dict(type='vqgan'))
    assert float(q.embedding.weight.detach().abs().max()) <= 1.0 / K + 1e-9           # VQGANQuantizer init: U(-1/K, 1/K)
    set_weight(q, w)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    memo = {}
    z, loss, memo = q(xd, memo)
    quant = memo['quant'].cpu().numpy()
    np.testing.assert_array_equal(quant, co.l2_argmin(x, w))
    assert memo['x'] is xd and set(memo['loss']) == {'vqgan_loss'}
    up = synth.normal(5, N, D)
    (loss + (z * torch.from_numpy(up).cuda()).sum()).backward()
    rl, rz, rgx, rgw = ref_grads(x, w, quant, upstream=torch.from_numpy(up))
    np.testing.assert_array_equal(z.detach().cpu().numpy(), rz)              # x + (z - x): bit-exact
    assert abs(loss.item() - rl) <= 1e-5 * max(1, abs(rl))
    assert abs(memo['loss']['vqgan_loss'].item() - rl) <= 1e-5 * max(1, abs(rl))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), rgx, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(q.embedding.weight.grad.cpu().numpy(), rgw, rtol=1e-4, atol=1e-7)
    # histogram from the fused epilogue and lazy distance
    assert 'hist' not in memo['encode']                                   # plain VQGAN forward: no histogram pass
    d = memo['encode']['distance'].materialize()
    assert d.requires_grad                                               # autograd reaches x and the codebook through it
    np.testing.assert_array_equal(d.detach().cpu().numpy(), co.l2_dist(x, w))


def test_bf16_latents_under_autocast_semantics():
    N, K, D = 512, 16384, 256
    x, w = synth.make_inputs('normal_bf16x', 3407, N, K, D)
    q = build(vqgan_cfg(K, D), train=False)
    set_weight(q, w)
    xd = torch.from_numpy(x).cuda().bfloat16()
    z, loss, memo = q(xd, {})
    gz = np.load(os.path.join(GOLDEN, 'l2_c2_bf16x_s3407.npz'))
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), gz['quant'].astype(np.int64))
    assert z.dtype == torch.float32 and abs(loss.item() - float(gz['loss'])) <= 1e-5 * max(1, float(gz['loss']))
    # tokenize-only entry point (models/base.py:141) and decode of an image-shaped index tensor (models.py:102)
    x2, quant, _ = q.encode(xd, {})
    assert x2 is xd and torch.equal(quant, memo['quant'])
    zz, _ = q.decode(quant.reshape(2, 16, 16), {})
    assert zz.shape == (2, 16, 16, D)
    np.testing.assert_array_equal(zz.detach().reshape(-1, D).cpu().numpy(), w[gz['quant'].astype(np.int64)])


def test_normalize_callback_llamagen_shape():
    # configs/llamagen/vqgan.py: D=8, NormalizeCallback, L2
    N, K, D = 2048, 16384, 8
    x, w = synth.make_inputs('normal', 3407, N, K, D)
    q = build(vqgan_cfg(K, D, callbacks=[dict(type='NormalizeCallback')]), train=False)
    set_weight(q, w)
    z, loss, memo = q(torch.from_numpy(x).cuda(), {})
    xo, wo = co.normalize_rows(x), co.normalize_rows(w)
    np.testing.assert_array_equal(memo['x'].cpu().numpy(), xo)                # x is replaced by normalize(x)
    np.testing.assert_array_equal(q.embedding.weight.detach().cpu().numpy(), wo)   # weight.data rebound, even in eval
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), co.l2_argmin(xo, wo))
    gz = np.load(os.path.join(GOLDEN, 'norml2_llamagen_d8.npz'))
    assert abs(loss.item() - float(gz['loss'])) <= 1e-5
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), gz['quant'].astype(np.int64))   # vs the reference's ATen ops: 0 rows differ


def vqkd_cfg(K, D):
    # configs/vqkd/model.py:20-26
    return dict(type='VQKDQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                distance=dict(type='CosineDistance'), callbacks=[dict(type='VQKDCallback', ema=dict())],
                losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))


def test_vqkd_train_step_matches_golden():
    g = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    q = build(vqkd_cfg(K, D), train=True)
    set_weight(q, w)
    q._forward_pre_hooks.clear()              # skip the k-means lazy init: the fixture starts from a given codebook
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    z, loss, memo = q(xd, {})
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g['quant'].astype(np.int64))
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g['w_new'], rtol=0, atol=3e-6)
    # loss = mse(normalize(z.detach()), normalize(x)) with z gathered from the UPDATED codebook, grads reach x only
    loss.backward()
    xt = torch.from_numpy(x).requires_grad_(True)
    xn = torch.nn.functional.normalize(xt)
    zt = tr.decode(torch.from_numpy(g['quant'].astype(np.int64)), torch.from_numpy(g['w_new']))
    rl = tr.commitment_loss(zt, xn, norm=True)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 1e-5
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-8)
    assert q.embedding.weight.grad is None
    # eval mode: no update (callbacks.py:121-122) apart from the idempotent-valued normalisation
    q.eval()
    before = q.embedding.weight.detach().clone()
    q(torch.from_numpy(x).cuda(), {})
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), before.cpu().numpy(), rtol=0, atol=2e-7)


def cvq_cfg(K, D, dist='Cosine'):
    # configs/cvqvae/quantizer.py:1-6 on top of configs/vqgan/model.py:19-23 (configs/cvqvae/*_ddp.py: distance Cosine)
    return vqgan_cfg(K, D, dist, callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])


def test_unchanged_configs_follow_autocast():
    """north_star: "drops in under the existing VQGAN / VQ-KD / CVQ-VAE configs unchanged".  The reference's GPU trainers
    and validators run every step inside torch.autocast('cuda', bf16) (vq/runners/base.py:30-48), where CosineDistance's
    einsum is a bf16 matmul with a bf16 result (distances.py:39-46).  The SAME config dicts — no `autocast=` key — must
    return the reference's autocast results inside such a region and its fp32 results outside.  Fixtures: the reference's
    own modules under autocast (autocast_modules.npz, cosbf16_*.npz) and without (update_*.npz, cos_c3_unit_s3407.npz)."""
    g = np.load(os.path.join(GOLDEN, 'autocast_modules.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    xd = torch.from_numpy(x).cuda()
    ac = lambda: torch.autocast('cuda', dtype=torch.bfloat16)    # noqa: E731

    # ---- VQ-KD (configs/vqkd/model.py:20-26): eval step, then a train step --------------------------------------
    q = build(vqkd_cfg(K, D), train=False)
    q._forward_pre_hooks.clear()
    set_weight(q, w)
    with torch.no_grad(), ac():
        _, loss, memo = q(xd, {})
    assert memo['encode']['distance'].dtype == torch.bfloat16          # as the reference's matrix under autocast
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g['eval_quant'].astype(np.int64))
    assert abs(loss.item() - float(g['eval_loss'])) <= 1e-5
    fp32 = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    set_weight(q, w)
    with torch.no_grad():
        _, _, memo32 = q(xd, {})                                       # the very same module outside autocast: fp32 definition
    assert memo32['encode']['distance'].dtype == torch.float32
    np.testing.assert_array_equal(memo32['quant'].cpu().numpy(), fp32['quant'].astype(np.int64))
    assert int((memo32['quant'] != memo['quant']).sum()) == int(g['eval_differs_from_fp32']) > 0
    q.train()
    set_weight(q, w)
    xg = xd.clone().requires_grad_(True)
    with ac():
        _, loss, memo = q(xg, {})
    loss.backward()
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g['vqkd_quant'].astype(np.int64))
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g['vqkd_w_new'], rtol=0, atol=3e-6)
    assert abs(loss.item() - float(g['vqkd_loss'])) <= 1e-5
    np.testing.assert_allclose(xg.grad[:8].cpu().numpy(), g['vqkd_grad_x_head'], rtol=1e-4, atol=1e-8)

    # ---- CVQ-VAE (configs/cvqvae/quantizer.py:1-6), two train steps: row argmin AND NearestAnchor's column argmin ------
    q = build(cvq_cfg(K, D), train=True, init=dict(type='vqgan'))
    set_weight(q, w)
    for quant_key, col_key, p_key, w_key in (('cvq_quant', 'cvq_col_idx', 'cvq_p1', 'cvq_w_new'),
                                             ('cvq_quant2', 'cvq_col_idx2', 'cvq_p2', 'cvq_w_new2')):
        with torch.no_grad(), ac():
            _, _, memo = q(xd, {})
            col = memo['encode']['distance'].argmin(0)                 # as quantizer_callback.py:86 / anchors.py:83 read it
        np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g[quant_key].astype(np.int64))
        np.testing.assert_array_equal(col.cpu().numpy(), g[col_key].astype(np.int64))
        np.testing.assert_allclose(q.get_buffer('_probability').cpu().numpy(), g[p_key], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g[w_key], rtol=0, atol=3e-6)
    g32 = np.load(os.path.join(GOLDEN, 'update_cvq_cosine.npz'))
    q = build(cvq_cfg(K, D), train=True, init=dict(type='vqgan'))
    set_weight(q, w)
    with torch.no_grad():
        _, _, memo = q(xd, {})
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g32['quant'].astype(np.int64))
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g32['w_new'], rtol=0, atol=3e-6)

    # ---- BASELINE configs[2] (K = 8192, D = 32) against the CosineDistance fixtures of both modes; L2 is unaffected -----
    for name, inside in (('cosbf16_c3_unit', True), ('cos_c3_unit_s3407', False)):
        gz = np.load(os.path.join(GOLDEN, name + '.npz'))
        sp = json.loads(str(gz['spec']))
        x3, w3 = synth.make_inputs(sp['kind'], sp['seed'], sp['N'], sp['K'], sp['D'])
        q = build(dict(vqkd_cfg(sp['K'], sp['D']), callbacks=[]), train=False)
        set_weight(q, w3)
        with torch.no_grad():
            if inside:
                with ac():
                    _, _, memo = q(torch.from_numpy(x3).cuda(), {})
                    col = memo['encode']['distance'].argmin(0)
            else:
                _, _, memo = q(torch.from_numpy(x3).cuda(), {})
                col = memo['encode']['distance'].argmin(0)
        np.testing.assert_array_equal(memo['quant'].cpu().numpy(), gz['quant'].astype(np.int64))
        np.testing.assert_array_equal(col.cpu().numpy(), gz['col_idx'].astype(np.int64))
    gz = np.load(os.path.join(GOLDEN, 'l2_c2_bf16x_s3407.npz'))
    x2, w2 = synth.make_inputs('normal_bf16x', 3407, 512, 16384, 256)
    q = build(vqgan_cfg(16384, 256), train=False)
    set_weight(q, w2)
    with torch.no_grad(), ac():
        _, loss, memo = q(torch.from_numpy(x2).cuda().bfloat16(), {})      # the conv connector hands over bf16 latents
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), gz['quant'].astype(np.int64))
    assert memo['encode']['distance'].dtype == torch.float32 and abs(loss.item() - float(gz['loss'])) <= 1e-5
    # explicit overrides stay available
    from vector_quantization_amd import quantizers as Q
    with ac():
        assert Q.CosineDistance(autocast=None).metric == 'Cosine' and Q.CosineDistance().metric == 'CosineBF16'
    assert Q.CosineDistance(autocast='bf16').metric == 'CosineBF16' and Q.CosineDistance().metric == 'Cosine'


def test_vqkd_kmeans_lazy_init_runs():
    N, K, D = 4096, 256, 32
    x, _ = synth.make_inputs('normal', 41, N, K, D)
    q = build(vqkd_cfg(K, D), train=True)
    z, loss, memo = q(torch.from_numpy(x).cuda(), {})
    w = q.embedding.weight.detach()
    np.testing.assert_allclose(w.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    np.testing.assert_array_equal(memo['encode']['hist'].cpu().numpy().astype(np.int64),
                                  co.bincount(memo['quant'].cpu().numpy(), K))   # fused epilogue histogram (training callbacks)
    used = int((memo['encode']['hist'] > 0).sum())
    assert used > K // 2, f'k-means init left {K - used} dead codes'
    assert len(q._forward_pre_hooks) == 0       # the one-shot hook removed itself


@pytest.mark.parametrize('dist', ['L2', 'Cosine'])
def test_cvq_train_step_matches_golden(dist):
    g = np.load(os.path.join(GOLDEN, f'update_cvq_{dist.lower()}.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    cfg = vqgan_cfg(K, D, dist, callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])
    q = build(cfg, train=True, init=dict(type='vqgan'))                # configs/cvqvae/quantizer.py:1-6
    assert '_probability' in q.state_dict()
    set_weight(q, w)
    z, loss, memo = q(torch.from_numpy(x).cuda(), {})
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g['quant'].astype(np.int64))
    np.testing.assert_allclose(q.get_buffer('_probability').cpu().numpy(), g['p1'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g['w_new'], rtol=0, atol=3e-6)
    q.eval()
    before = q.embedding.weight.detach().clone()
    q(torch.from_numpy(x).cuda(), {})
    assert torch.equal(q.embedding.weight.detach(), before)               # quantizer_callback.py:82-83


def test_state_dict_layout_and_cached_codebook():
    K, D = 1024, 256
    q = build(vqgan_cfg(K, D, cache_codebook=True), train=False, init=dict(type='vqgan'))
    keys = set(q.state_dict())
    # tools/convert_checkpoints.py:195-199,239-243 (with the '_quantizer.' prefix stripped)
    assert keys == {'_embedding.weight', '_losses.vqgan_loss._weight._steps',
                    '_losses.vqgan_loss._codebook._weight._steps', '_losses.vqgan_loss._codebook._mse._weight._steps',
                    '_losses.vqgan_loss._commitment._weight._steps',
                    '_losses.vqgan_loss._commitment._mse._weight._steps'}
    x, w = synth.make_inputs('normal', 1, 300, K, D)
    set_weight(q, w)
    q.invalidate_codebook()
    xd = torch.from_numpy(x).cuda()
    q1 = q.encode(xd, {})[1]
    img = q._prepared
    q2 = q.encode(xd, {})[1]
    assert q._prepared is img and torch.equal(q1, q2)                       # image reused while weight is unchanged
    with torch.no_grad():
        q.embedding.weight.mul_(-1.0)                                        # in-place change bumps _version
    q3 = q.encode(xd, {})[1]
    assert q._prepared is not img
    np.testing.assert_array_equal(q3.cpu().numpy(), co.l2_argmin(x, -w))


@pytest.mark.gpu
def test_deterministic_mode_reproducible_codebook_gradient():
    """torch.use_deterministic_algorithms(True) routes the codebook-side sums through the ordered kernels: two
    identical training steps give bit-identical weight gradients (the atomic route differs in the last bits)."""
    from vector_quantization_amd import build_quantizer, Config
    N, K, D = 20000, 256, 64
    g = synth.rng(21)
    q = build_quantizer(dict(type='VQGANQuantizer',
                             embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
    q.init_weights(Config(type='vqgan'))
    q = q.cuda().train()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g.standard_normal((K, D), dtype=np.float32)))
    x = torch.from_numpy(g.standard_normal((N, D), dtype=np.float32)).cuda()
    prev = torch.are_deterministic_algorithms_enabled()
    grads = []
    try:
        torch.use_deterministic_algorithms(True)
        for _ in range(3):
            q.zero_grad()
            xi = x.clone().requires_grad_(True)
            z, loss, _ = q(xi, {})
            (loss + (z * z).mean()).backward()
            grads.append((q.embedding.weight.grad.clone(), xi.grad.clone()))
    finally:
        torch.use_deterministic_algorithms(prev)
    for gw, gx in grads[1:]:
        assert torch.equal(gw, grads[0][0]) and torch.equal(gx, grads[0][1])
    q.zero_grad()
    xi = x.clone().requires_grad_(True)
    z, loss, _ = q(xi, {})
    (loss + (z * z).mean()).backward()                  # default policy at this size: atomics
    assert torch.allclose(q.embedding.weight.grad, grads[0][0], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('dist', ['L2', 'Cosine'])
def test_cvq_sparse_anchor_exchange_same_result(dist):
    """CVQVAECallback's default data flow — anchors only for the codes whose decay can come out below 1, listed on the
    device (vqhip_cvq_rows), column argmin over those (vqhip_col_argmin_rows), one fused update (vqhip_cvq_apply) — against
    the reference's dense flow (sparse_anchors=False): bit-identical codebooks, probabilities and tokens step after step,
    a list that shrinks as codes come into use, and no host synchronisation after the first step."""
    N, K, D = 3000, 2048, 64
    g = synth.rng(31)
    w0 = synth.unit_rows(g.standard_normal((K, D), dtype=np.float32))
    xs = [g.standard_normal((N, D), dtype=np.float32) * np.float32(0.3) + w0[g.integers(0, K // 8, N)] for _ in range(6)]
    outs = []
    for sparse in (False, None):
        cfg = vqgan_cfg(K, D, dist, callbacks=[dict(type='CVQVAECallback', ema=dict(), sparse_anchors=sparse,
                                                    anchor=dict(type='NearestAnchor'))])
        q = build(cfg, train=True, init=dict(type='vqgan'))
        set_weight(q, w0)
        quants, rows = [], []
        for x in xs:                                     # most tokens sit on an eighth of the codes: the rest go stale
            _, _, memo = q(torch.from_numpy(x).cuda(), {})
            quants.append(memo['quant'].clone())
            rows.append(q._callbacks.callbacks[0].last_exchange_rows)
        outs.append((q.embedding.weight.detach().clone(), q.get_buffer('_probability').clone(), quants, rows))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
    rows = outs[1][3]
    assert outs[0][3] == [None] * 6 and rows[0] == K and rows[-1] < rows[0], rows
    from vector_quantization_amd import ops
    # the list is a superset of {decay < 1} for the step it serves, and tight: nothing with decay == 1 far inside
    p_prev = outs[1][1].contiguous()
    r, slot, count = ops.cvq_rows(p_prev, K, 0.99, 1e-3)
    listed = torch.zeros(K, dtype=torch.bool, device='cuda')
    listed[r[: int(count)].long()] = True
    assert torch.equal(slot >= 0, listed) and torch.equal(r[: int(count)].long(), torch.nonzero(listed).reshape(-1))
    for freq in (torch.zeros(K, device='cuda'), torch.rand(K, device='cuda') / K):
        p_next = p_prev * 0.99 + freq * (1 - 0.99)
        decay = ops.cvq_decay(p_next.contiguous(), K, 0.99, 1e-3)
        assert bool((decay[~listed] == 1.0).all())
    assert 0 < int(count) < K


@pytest.mark.parametrize('scale', [1.0, 3.0, 40.0, 1e4])
def test_column_pass_under_cosine_with_rows_that_are_not_unit_norm(scale):
    """NearestAnchor's role-swapped pass runs on the operands as given (include/vqhip.h: "COS: x and e already normalised") and
    builds its fp16 image in one launch with the scale of unit rows (2^13).  Rows that break the contract must still give
    the fp32 definition 1 - x.e bit for bit: moderately large values keep rigorous margins (the tile maxima come from the data),
    values beyond fp16 range at that scale raise the non-finite flag and every row takes the fp32 pass."""
    from vector_quantization_amd import ops
    g = torch.Generator(device='cuda').manual_seed(int(scale) + 11)
    N, K, D = 1500, 1024, 64
    x = torch.randn(N, D, device='cuda', generator=g) * scale
    w = torch.randn(K, D, device='cuda', generator=g)
    d = ops.distance(x, w, 'Cosine')
    rows_n = torch.arange(N, device='cuda')[:, None].expand(N, K)
    want = torch.where(d == d.min(0, keepdim=True).values, rows_n, N).min(0).values
    assert torch.equal(ops.col_argmin(x, w, 'Cosine'), want)
    rows = torch.arange(0, K, 3, device='cuda', dtype=torch.int32)
    count = torch.tensor([rows.numel()], device='cuda', dtype=torch.int32)
    sub = ops.col_argmin_rows(x, w, rows, count, rows.numel() + 5, 'Cosine')
    assert torch.equal(sub[:rows.numel()], want[rows.long()])


def test_sparse_anchor_pieces_match_the_dense_ops():
    """vqhip_col_argmin_rows == vqhip_col_argmin on the listed codes (every metric, bf16 latents, a capacity above the
    count), the packed count header survives a float SUM exactly, and vqhip_cvq_apply == vqhip_cvq_step."""
    from vector_quantization_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    N, K, D = 3072, 4096, 64
    x = torch.randn(N, D, device='cuda', generator=g)
    w = torch.randn(K, D, device='cuda', generator=g)
    p = torch.rand(K, device='cuda', generator=g) * 4e-5            # decay arguments straddle the threshold
    rows, slot, count = ops.cvq_rows(p, K, 0.99, 1e-3)
    M = int(count)
    assert 0 < M < K
    for metric, xx, ww in (('L2', x, w), ('L2', x.bfloat16(), w), ('Cosine', ops.normalize_rows(x), ops.normalize_rows(w)),
                           ('CosineBF16', ops.normalize_rows(x).bfloat16().float(), ops.normalize_rows(w).bfloat16().float())):
        full = ops.col_argmin(xx, ww, metric)
        for cap in (M, K, min(K, M + 77)):
            sub = ops.col_argmin_rows(xx, ww, rows, count, cap, metric)
            assert torch.equal(sub[:M], full[rows[:M].long()]), (metric, cap)
    # empty list: nothing is launched for the column pass, the update still runs
    p_hot = torch.full((K,), 1e-3, device='cuda')
    r0, s0, c0 = ops.cvq_rows(p_hot, K, 0.99, 1e-3)
    assert int(c0) == 0 and bool((s0 == -1).all())
    # packed counts: 3 "ranks" summed in fp32, counts far beyond 2^24 in total
    hists = [torch.randint(0, 2 ** 31 - 1, (K,), device='cuda', dtype=torch.int64, generator=g).to(torch.int32) for _ in range(3)]
    numels = [2 ** 31 - 5, 12345, 2 ** 40 + 17]
    acc = torch.zeros(ops.pack_floats(K, 0, D), device='cuda')
    for h, n in zip(hists, numels):
        buf = torch.empty_like(acc)
        ops.pack_counts(h, n, buf)
        assert float(buf[:2 * K + 3].max()) < 65536 or n >= 2 ** 32
        acc += buf
    out = ops.unpack_counts(acc, K)
    assert torch.equal(out[:K], sum(h.to(torch.int64) for h in hists)) and int(out[K]) == sum(numels)
    h64 = hists[0].to(torch.int64) * 3
    buf = torch.empty_like(acc)
    ops.pack_counts(h64, 7, buf)
    assert torch.equal(ops.unpack_counts(buf, K)[:K], h64)
    # one-rank apply == cvq_step (dense) given the same column indices
    hist = torch.bincount(torch.randint(0, K, (N,), device='cuda', generator=g), minlength=K).to(torch.int32)
    col_full = ops.col_argmin(x, w, 'L2')
    w_a, p_a = torch.empty_like(w), torch.empty_like(p)
    ops.cvq_step(w, w_a, p, p_a, hist, N, x, col_full, 0.99, 1e-3)
    w_b, p_b = torch.empty_like(w), torch.empty_like(p)
    ops.cvq_apply(w, w_b, p, p_b, slot, 0.99, 1e-3, hist32=hist, numel=N, x=x, col_idx=ops.col_argmin_rows(x, w, rows, count, K, 'L2'))
    assert torch.equal(w_a, w_b) and torch.equal(p_a, p_b)
    # packed apply with world = 1 payload == the same
    packed = ops.cvq_pack(hist, N, x, col_full[rows[:M].long()].contiguous(), count, M, K)
    assert packed.numel() == 2 * K + 4 + M * D
    w_c, p_c = torch.empty_like(w), torch.empty_like(p)
    ops.cvq_apply(w, w_c, p, p_c, slot, 0.99, 1e-3, packed=packed, world=1)
    assert torch.equal(w_a, w_c) and torch.equal(p_a, p_c)


def test_graphed_quantizer_follows_weight_changes():
    """a GraphedQuantizer built on a quantizer with cache_codebook=True used to replay the codebook image frozen
    at capture.  The constructor now switches the cache off; a weight changed between replays (optimizer step,
    load_state_dict) is what the next replay quantizes against."""
    from vector_quantization_amd.graphs import GraphedQuantizer
    N, K, D = 2048, 1024, 64
    gen = synth.rng(23)
    w0 = gen.standard_normal((K, D), dtype=np.float32)
    w1 = gen.standard_normal((K, D), dtype=np.float32)
    x = gen.standard_normal((N, D), dtype=np.float32)
    q = build(vqgan_cfg(K, D, cache_codebook=True), train=False, init=dict(type='vqgan'))
    set_weight(q, w0)
    xd = torch.from_numpy(x).cuda()
    gq = GraphedQuantizer(q, xd)
    assert q._cache_codebook is False
    _, _, quant = gq(xd)
    np.testing.assert_array_equal(quant.cpu().numpy(), co.l2_argmin(x, w0))
    set_weight(q, w1)                                    # in-place change of the live parameter
    _, _, quant = gq(xd)
    np.testing.assert_array_equal(quant.cpu().numpy(), co.l2_argmin(x, w1))
    q.load_state_dict({k: (torch.from_numpy(w0).cuda() if k == '_embedding.weight' else v) for k, v in q.state_dict().items()})
    z, loss, quant = gq(xd)
    np.testing.assert_array_equal(quant.cpu().numpy(), co.l2_argmin(x, w0))
    np.testing.assert_array_equal(z.cpu().numpy(), co.gather_ste(x, w0, co.l2_argmin(x, w0))[1])


# ---- rows that were "partial" in round 1: EntropyLoss with autograd, alternative anchors, k-means lazy init --------

@pytest.mark.parametrize('dist', ['L2', 'Cosine'])
def test_entropy_loss_value_and_gradients_match_reference(dist):
    """EntropyLoss (losses.py:130-153) over memo['encode']['distance'] — a LazyDistance materialised by the HIP distance
    kernel with autograd — against the fixture produced by the reference's own EntropyLoss + L2/CosineDistance."""
    g = np.load(os.path.join(GOLDEN, 'entropy_loss.npz'))
    spec = json.loads(str(g['spec']))
    x, w = g['x'], g['w']
    K, D = w.shape
    cfg = vqgan_cfg(K, D, dist)
    cfg['losses']['entropy'] = dict(type='EntropyLoss', temperature=spec['temperature'])
    q = build(cfg, train=True, init=dict(type='vqgan'))
    set_weight(q, w)
    assert not q._fusable()
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    memo = {}
    x2, quant, memo = q.encode(xd, memo)
    d = memo['encode']['distance']
    assert isinstance(d, torch.Tensor) and d.shape == (x.shape[0], K)
    val = q._losses['entropy'](None, x2, dict(distance=d))            # the reference reads memo['distance'] (losses.py:145)
    key = dist.lower()
    assert abs(val.item() - float(g[f'loss_{key}'])) <= 1e-5 * max(1.0, abs(float(g[f'loss_{key}'])))
    val.backward()
    gx, gw = xd.grad.cpu().numpy(), q.embedding.weight.grad.cpu().numpy()
    scale_x, scale_w = np.abs(g[f'grad_x_{key}']).max(), np.abs(g[f'grad_w_{key}']).max()
    np.testing.assert_allclose(gx, g[f'grad_x_{key}'], rtol=1e-4, atol=1e-5 * scale_x)
    np.testing.assert_allclose(gw, g[f'grad_w_{key}'], rtol=1e-4, atol=1e-5 * scale_w)


def test_distance_matrix_backward_matches_torch():
    """L2Distance / CosineDistance forward (HIP) + backward (two GEMMs) against torch.cdist / the einsum definition."""
    from vector_quantization_amd import quantizers as Q
    N, K, D = 300, 200, 64
    x, w = synth.make_inputs('normal', 77, N, K, D)
    up = synth.normal(78, N, K)
    for cls, ref in ((Q.L2Distance, tr.l2_distance), (Q.CosineDistance, tr.cosine_distance)):
        xd = torch.from_numpy(x).cuda().requires_grad_(True)
        wd = torch.from_numpy(w).cuda().requires_grad_(True)
        d = cls()(xd, wd)
        (d * torch.from_numpy(up).cuda()).sum().backward()
        xt = torch.from_numpy(x).requires_grad_(True)
        wt = torch.from_numpy(w).requires_grad_(True)
        dr = ref(xt, wt)
        (dr * torch.from_numpy(up)).sum().backward()
        np.testing.assert_allclose(d.detach().cpu().numpy(), dr.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(wd.grad.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_multinomial_and_cached_anchor_match_reference():
    """MultinomialAnchor: the sampling distribution equals the reference's softmax over each code's column (the draw
    itself uses the device generator); CachedAnchor: host-side draws => the very rows the reference picked."""
    import random

    from vector_quantization_amd import quantizers as Q
    g = np.load(os.path.join(GOLDEN, 'anchors_alt.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    assert synth.sha(x) == str(g['x_sha']) and synth.sha(w) == str(g['w_sha'])
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    dist = Q.L2Distance()
    d = Q.LazyDistance(dist, xd, wd)
    quant = d.argmin(-1)
    p = torch.zeros(K, device='cuda')
    ma = Q.MultinomialAnchor()
    np.testing.assert_allclose(ma.probabilities(d).cpu().numpy(), g['multinomial_probs'], rtol=2e-5, atol=1e-8)
    torch.manual_seed(7)
    a, _ = ma(xd, wd, d, quant, p)
    assert a.shape == (K, D)
    rows = {r.tobytes() for r in x}
    assert all(r.tobytes() in rows for r in a.cpu().numpy())                      # every anchor is one of the latents
    ca = Q.CachedAnchor()
    random.seed(spec['python_seed_big'])
    a1, _ = ca(xd, wd, d, quant, p)                                                # N > K: random.sample
    np.testing.assert_array_equal(a1.cpu().numpy(), g['cached_big'])
    assert torch.equal(ca.cache, a1)
    torch.manual_seed(spec['torch_seed_eq'])
    a2, _ = ca(xd[:K], wd, Q.LazyDistance(dist, xd[:K], wd), quant[:K], p)         # N == K: CPU randperm
    np.testing.assert_array_equal(a2.cpu().numpy(), g['cached_eq'])
    torch.manual_seed(int(g['cached_small_seed']))
    a3, _ = ca(xd[:40], wd, Q.LazyDistance(dist, xd[:40], wd), quant[:40], p)      # N < K, cache tops the pool up
    np.testing.assert_array_equal(a3.cpu().numpy(), g['cached_small_with_cache'])
    sd = ca.state_dict()
    assert sd['_cache'].shape == (K, D)
    fresh = Q.CachedAnchor()
    fresh.load_state_dict(sd)                                                      # empty buffer resized on load
    assert torch.equal(fresh.cache.cpu(), ca.cache.cpu())


def test_vqkd_lazy_init_matches_oracle_and_reference():
    """f4: the 10 on-device Lloyd iterations of VQKDCallback.lazy_init_weights from the seeded random.sample start.
    Every iteration's assignment is the exact argmin for the codebook of that iteration (C oracle, bit-exact), the
    centroid update equals the oracle's, and the whole run reproduces the reference's (fixture from its own
    lazy_init_weights): same start rows, per-iteration assignments equal except fp32-envelope rows, final codebook 1e-5."""
    import random
    g = np.load(os.path.join(GOLDEN, 'lazy_init_vqkd.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, _ = synth.make_inputs('normal', spec['x_seed'], N, K, D)
    assert synth.sha(x) == str(g['x_sha'])
    q = build(vqkd_cfg(K, D), train=True)
    seen = []
    inner = q._encode

    def spy(xx, memo):
        quant, memo = inner(xx, memo)
        seen.append((quant.clone(), q.embedding.weight.detach().clone(), xx.detach().clone()))
        return quant, memo

    q._encode = spy
    after_init = []
    q.register_forward_pre_hook(lambda m, a: after_init.append(m.embedding.weight.detach().clone()))
    random.seed(spec['seed'])
    with torch.no_grad():
        q(torch.from_numpy(x).cuda(), {})
    assert len(seen) == spec['iters'] + 1 and len(q._forward_pre_hooks) == 1     # lazy hook removed itself, ours remains
    xn = co.normalize_rows(x)
    np.testing.assert_array_equal(seen[0][1].cpu().numpy(), co.normalize_rows(xn[g['indices']]))   # start = sampled rows
    flips = 0
    for it in range(spec['iters']):
        quant, book, xx = (t.cpu().numpy() for t in seen[it])
        np.testing.assert_array_equal(xx, xn)
        np.testing.assert_array_equal(quant, co.cos_argmin(xn, book))             # exact argmin for THIS codebook
        nxt = seen[it + 1][1].cpu().numpy() if it + 1 < spec['iters'] else after_init[0].cpu().numpy()
        want = co.normalize_rows(co.kmeans_centroids(xn, quant, book))
        np.testing.assert_allclose(nxt, want, rtol=0, atol=2e-6)                    # centroid update (atomic sum order)
        flips += int((quant != g['quants'][it].astype(np.int64)).sum())
    assert flips <= 1e-3 * N * spec['iters'], flips                                # vs the reference's own run
    np.testing.assert_allclose(after_init[0].cpu().numpy(), g['w_init'], rtol=0, atol=1e-5 if flips == 0 else 5e-2)


def test_cvq_full_size_c4_step():
    """BASELINE configs[3] at full size: K=16384, D=256, per-rank N=3072, cosine, CVQ-VAE train step through the module.
    Row indices vs the reference fixture (cos_c4_k16384), column argmin vs the fixture's d.argmin(0), and the update
    against the oracle formulas on those inputs."""
    g = np.load(os.path.join(GOLDEN, 'cos_c4_k16384_s3407.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs(spec['kind'], spec['seed'], N, K, D)
    cfg = vqgan_cfg(K, D, 'Cosine', callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])
    q = build(cfg, train=True, init=dict(type='vqgan'))
    set_weight(q, w)
    xd = torch.from_numpy(x).cuda()
    z, loss, memo = q(xd, {})
    quant = memo['quant'].cpu().numpy()
    np.testing.assert_array_equal(quant, g['quant'].astype(np.int64))
    col = memo['encode']['distance'].argmin(0).cpu().numpy()                      # the weights behind it are encode-time ones
    np.testing.assert_array_equal(col, g['col_idx'].astype(np.int64))
    p1 = co.ema(np.zeros(K, np.float32), (co.bincount(quant, K) / np.int64(N)).astype(np.float32), 0.99)
    np.testing.assert_allclose(q.get_buffer('_probability').cpu().numpy(), p1, rtol=1e-6, atol=1e-9)
    w_new = co.ema(w, x[col], co.cvq_decay(p1, K, 0.99, 1e-3))
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), w_new, rtol=0, atol=3e-6)


@pytest.mark.parametrize('mode', ['cvq_train', 'vqgan_eval'])
def test_graphed_quantizer_steps_match_eager(mode):
    """GraphedQuantizer (graphs.py): the whole module step — callbacks with the in-place codebook update, decode, loss and,
    in train mode, the backward — replayed from HIP graphs gives the eager module's results step after step: identical
    tokens, outputs and codebooks, gradients to summation-order tolerance."""
    from vector_quantization_amd.graphs import GraphedQuantizer
    N, K, D = 3072, 2048, 64
    gen = synth.rng(17)
    w0 = synth.unit_rows(gen.standard_normal((K, D), dtype=np.float32))
    xs = [gen.standard_normal((N, D), dtype=np.float32) for _ in range(3)]
    up = torch.from_numpy(gen.standard_normal((N, D), dtype=np.float32)).cuda()
    train = mode == 'cvq_train'
    cbs = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))] if train else []

    def make():
        q = build(vqgan_cfg(K, D, 'Cosine' if train else 'L2', callbacks=cbs), train=train, init=dict(type='vqgan'))
        set_weight(q, w0)
        return q

    def run(step):
        out = []
        for x in xs:
            xd = torch.from_numpy(x).cuda().requires_grad_(train)
            z, loss, quant = step(xd)
            if train:
                q_ = step.quantizer if hasattr(step, 'quantizer') else step.__self__
                q_.zero_grad(set_to_none=True)
                (loss + (z * up).sum()).backward()
                grads = (xd.grad.clone(), q_.embedding.weight.grad.clone())
            else:
                grads = None
            out.append((quant.clone(), z.detach().clone(), loss.detach().clone(), grads))
        return out

    class Eager:
        def __init__(self, q):
            self.quantizer = q

        def __call__(self, x):
            z, loss, memo = self.quantizer(x, {})
            return z, loss, memo['quant']

    q1, q2 = make(), make()
    ref = run(Eager(q1))
    gq = GraphedQuantizer(q2, torch.from_numpy(xs[0]).cuda())
    assert torch.equal(q2.embedding.weight.detach().cpu(), torch.from_numpy(w0))          # capture left no trace
    got = run(gq)
    for (qa, za, la, ga), (qb, zb, lb, gb) in zip(ref, got):
        assert torch.equal(qa, qb) and torch.equal(za, zb)
        assert abs(la.item() - lb.item()) <= 1e-6 * max(1.0, abs(la.item()))
        if train:
            np.testing.assert_allclose(gb[0].cpu().numpy(), ga[0].cpu().numpy(), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(gb[1].cpu().numpy(), ga[1].cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert torch.equal(q1.embedding.weight.detach(), q2.embedding.weight.detach())          # bit-identical codebooks after 3 steps
    if train:
        assert torch.equal(q1.get_buffer('_probability'), q2.get_buffer('_probability'))
        assert not torch.equal(q2.embedding.weight.detach().cpu(), torch.from_numpy(w0))   # and they did move


# ---- the benchmarked shapes themselves (BASELINE.json configs[1] and configs[4] at the size bench.py times) ------------

def test_full_size_bench_shape():
    """524 288 x 16384 x 256, bf16 latents, through VQGANQuantizer.forward in eval mode — exactly the step bench.py times:
    every row against the all-fp32 route (itself bit-equal to the C oracle on every fixture), 256 evenly spaced rows against
    the C oracle directly, the straight-through output literally x + (z - x) on a row sample, the loss against float64."""
    from vector_quantization_amd import ops
    N, K, D = 2048 * 256, 16384, 256
    g = torch.Generator(device='cuda').manual_seed(3407)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
    q = build(vqgan_cfg(K, D), train=False, init=dict(type='vqgan'))
    with torch.no_grad():
        q.embedding.weight.copy_(w)
        z, loss, memo = q(x, {})
    quant = memo['quant'].reshape(-1)
    exact = ops.argmin_exact(x, w, 'L2')
    assert int((quant != exact).sum()) == 0
    rows = torch.linspace(0, N - 1, 256, device='cuda').long()
    ref = co.l2_argmin(x[rows].float().cpu().numpy(), w.cpu().numpy())
    np.testing.assert_array_equal(quant[rows].cpu().numpy(), ref)
    xs, zs = x[rows].float(), w[quant[rows]]
    assert torch.equal(z[rows].float(), xs + (zs - xs))                                  # utils/ste.py:10, literally
    loss64 = 1.25 * float(((w[exact].double() - x.double()) ** 2).mean())                # VQGANLoss: m + 0.25 m (losses.py:73,127)
    assert abs(loss.item() - loss64) <= 1e-5 * max(1.0, abs(loss64))
    assert int(ops.hist(quant, K).sum()) == N


def test_full_size_tokenizer_shape():
    """BASELINE configs[4] at the metric's size: 2048 images = 524 288 tokens, K = 16384, D = 8, NormalizeCallback + L2
    (configs/llamagen/vqgan.py:10-20) through VQGANQuantizer.encode: every row against the all-fp32 route on the normalised
    operands, evenly spaced rows against the C oracle, x' = F.normalize(x) returned (quantizers/base.py:123-131)."""
    from vector_quantization_amd import ops
    N, K, D = 2048 * 256, 16384, 8
    g = torch.Generator(device='cuda').manual_seed(3407)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
    q = build(vqgan_cfg(K, D, callbacks=[dict(type='NormalizeCallback')]), train=False, init=dict(type='vqgan'))
    with torch.no_grad():
        q.embedding.weight.copy_(w)
        xn_out, quant, memo = q.encode(x, {})
    xn, wn = ops.normalize_rows(x), ops.normalize_rows(w)
    assert torch.equal(xn_out.float(), xn) and torch.equal(q.embedding.weight.detach(), wn)
    quant = quant.reshape(-1)
    assert int((quant != ops.argmin_exact(xn, wn, 'L2')).sum()) == 0
    rows = torch.linspace(0, N - 1, 256, device='cuda').long()
    xo, wo = co.normalize_rows(x[rows].float().cpu().numpy()), co.normalize_rows(w.cpu().numpy())
    np.testing.assert_array_equal(quant[rows].cpu().numpy(), co.l2_argmin(xo, wo))
