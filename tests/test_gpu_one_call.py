"""vqhip_cvq_forward / vqhip_vqkd_forward (include/vqhip.h, round 5): ONE host call per training forward.  The reference
runs the same step hook by hook (vq/algorithms/cvqvae/quantizer_callback.py:75-105, vq/algorithms/vqkd/quantizers/
callbacks.py:114-129); the golden fixtures produced by the reference's own modules (update_cvq_*.npz, update_vqkd.npz) are
checked through the one-call route in tests/test_gpu_modules.py (it is the default).  Here: the one-call route against the
hook-by-hook route of this package step after step — tokens, codebooks, probabilities and straight-through outputs bit for
bit, losses and gradients within 1e-6 — eager, under bf16 autocast, with in-place updates, and with the exchange forced
through its two-phase form."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, torch_ref as tr

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
EMB = 'torch_nn_modules_sparse_Embedding'


def build(cfg, w, no_grad_params=False):
    from vector_quantization_amd import Config, build_quantizer
    q = build_quantizer(cfg)
    q.train(True)
    q.init_weights(Config(type='vqgan') if cfg['type'] == 'VQGANQuantizer' else Config())
    q = q.cuda()
    q._forward_pre_hooks.clear()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(w))
    if no_grad_params:
        for p in q.parameters():
            p.requires_grad_(False)
    return q


def cvq_cfg(K, D, dist, loss='VQGANLoss'):
    return dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                distance=dict(type=f'{dist}Distance'), losses=dict(vqgan_loss=dict(type=loss)),
                callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))])


def vqkd_cfg(K, D):
    return dict(type='VQKDQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                distance=dict(type='CosineDistance'), callbacks=[dict(type='VQKDCallback', ema=dict())],
                losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))


def batches(N, K, D, w0, steps, seed, bf16=False):
    g = synth.rng(seed)
    out = []
    for _ in range(steps):                       # most tokens sit on an eighth of the codes: the rest go stale and get listed
        x = g.standard_normal((N, D), dtype=np.float32) * np.float32(0.3) + w0[g.integers(0, max(1, K // 8), N)]
        t = torch.from_numpy(x).cuda()
        out.append(t.bfloat16() if bf16 else t)
    return out


def run_steps(q, xs, gz, one_call, autocast=False):
    q.one_call_steps = one_call
    rec = []
    for x in xs:
        xin = x.clone().requires_grad_(True)
        for p in q.parameters():
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            z, loss, memo = q(xin, {})
        torch.autograd.backward([loss, z], [None, gz])
        wg = q.embedding.weight.grad
        rec.append(dict(quant=memo['quant'].clone(), z=z.detach().clone(), loss=loss.detach().clone(), gx=xin.grad.clone(),
                        gw=None if wg is None else wg.clone(), w=q.embedding.weight.detach().clone(),
                        hist=memo['encode']['hist'].clone(), memo_x=memo['x'].detach().clone(),
                        loss_memo={k: v.detach().clone() for k, v in memo['loss'].items()}))
    return rec


def assert_same(a, b, exact_w=True, tol=1e-6):
    for sa, sb in zip(a, b):
        assert torch.equal(sa['quant'], sb['quant'])
        assert torch.equal(sa['hist'], sb['hist'])
        assert torch.equal(sa['memo_x'], sb['memo_x'])
        if exact_w:
            assert torch.equal(sa['w'], sb['w']) and torch.equal(sa['z'], sb['z'])
        else:
            torch.testing.assert_close(sa['w'], sb['w'], rtol=0, atol=3e-6)
            torch.testing.assert_close(sa['z'], sb['z'], rtol=0, atol=3e-6)
        assert abs(float(sa['loss']) - float(sb['loss'])) <= tol * max(1.0, abs(float(sb['loss'])))
        assert sa['loss_memo'].keys() == sb['loss_memo'].keys()
        torch.testing.assert_close(sa['gx'], sb['gx'], rtol=1e-5, atol=1e-9)
        assert (sa['gw'] is None) == (sb['gw'] is None)
        if sa['gw'] is not None:
            torch.testing.assert_close(sa['gw'], sb['gw'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('dist,D,bf16', [('L2', 64, False), ('Cosine', 64, False), ('Cosine', 256, True), ('L2', 8, True)])
def test_cvq_one_call_equals_hook_by_hook(dist, D, bf16):
    N, K = 3000, 2048
    w0 = synth.unit_rows(synth.rng(5).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 6, 31, bf16)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1)) / (N * D)
    recs, rows = [], []
    for one_call in (False, True):
        q = build(cvq_cfg(K, D, dist), w0)
        recs.append(run_steps(q, xs, gz, one_call))
        rows.append(q._callbacks.callbacks[0].last_exchange_rows)
        recs[-1].append(dict(p=q.get_buffer('_probability').clone()))
    assert torch.equal(recs[0][-1]['p'], recs[1][-1]['p'])
    assert_same(recs[0][:-1], recs[1][:-1])
    assert rows[0] == rows[1] and 0 < rows[1] < K, rows           # the one-call route sized its launches from the prefetched count


def test_cvq_one_call_under_autocast_and_other_losses():
    N, K, D = 2048, 1024, 32
    w0 = synth.unit_rows(synth.rng(6).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 4, 32)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(2)) / (N * D)
    for loss in ('VQGANLoss', 'CodebookLoss'):                    # configs/cluster/model.py: CodebookLoss
        recs = []
        for one_call in (False, True):
            q = build(cvq_cfg(K, D, 'Cosine', loss), w0)
            recs.append(run_steps(q, xs, gz, one_call, autocast=True))
        assert_same(recs[0], recs[1])


def test_cvq_one_call_in_place_updates_and_external_probability_change():
    """inplace_updates (what graph capture uses) writes codebook and probabilities into their storage; a probability buffer
    replaced from outside (a loaded checkpoint) voids the prefetched list and is counted on the spot."""
    N, K, D = 1500, 1024, 64
    w0 = synth.unit_rows(synth.rng(7).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 5, 33)
    gz = torch.zeros(N, D, device='cuda')
    outs = []
    for one_call, inplace in ((False, False), (True, True), (True, False)):
        q = build(cvq_cfg(K, D, 'L2'), w0)
        q.inplace_updates = inplace
        ptr = q.embedding.weight.data_ptr()
        rec = run_steps(q, xs[:3], gz, one_call)
        if inplace:
            assert q.embedding.weight.data_ptr() == ptr
        sd = {k: v.clone() for k, v in q.state_dict().items()}
        q.load_state_dict(sd)                                     # in-place copy_: bumps the buffer's version
        with torch.no_grad():
            q.get_buffer('_probability').mul_(0.5)                # ... and a change of values from outside
        rec += run_steps(q, xs[3:], gz, one_call)
        outs.append((rec, q.get_buffer('_probability').clone()))
    for other in outs[1:]:
        assert_same(outs[0][0], other[0])
        assert torch.equal(outs[0][1], other[1])


# 40 000 tokens: the ordered (bit-reproducible) centroid sums; D = 8 / 16 / 24: the other widths of the L-lanes-per-row front, tail
# and backward kernels (vqhip_step_kernels.h: 8, 16 and 32 lanes per row, D = 24 leaves lanes of a group idle); D = 64: the wave-per-row forms
@pytest.mark.parametrize('N,D', [(3000, 32), (40000, 32), (3000, 8), (3000, 16), (3000, 24), (3000, 64)])
def test_vqkd_one_call_equals_hook_by_hook(N, D):
    K = 1024
    w0 = synth.unit_rows(synth.rng(8).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 4, 34)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) / (N * D)
    recs = []
    for one_call in (False, True):
        q = build(vqkd_cfg(K, D), w0, no_grad_params=True)
        recs.append(run_steps(q, xs, gz, one_call))
    assert_same(recs[0], recs[1], exact_w=(N >= 32768))
    # autocast: the bf16 cosine metric inside the one call as well
    recs = []
    for one_call in (False, True):
        q = build(vqkd_cfg(K, D), w0, no_grad_params=True)
        recs.append(run_steps(q, xs[:2], gz, one_call, autocast=True))
    assert_same(recs[0], recs[1], exact_w=(N >= 32768))


def test_vqkd_one_call_loss_is_reproducible_bit_for_bit():
    """The loss of the one-call VQ-KD forward is a fixed-order sum of per-workgroup partials (vqkd_tail_finish): the same step
    from the same state gives the same bits (above 32 768 tokens the centroid sums are the ordered ones, so the codebooks are too)."""
    K, D, N = 1024, 32, 40000
    w0 = synth.unit_rows(synth.rng(8).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 2, 35)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) / (N * D)
    runs = [run_steps(build(vqkd_cfg(K, D), w0, no_grad_params=True), xs, gz, True) for _ in range(3)]
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a['loss'], b['loss']) and torch.equal(a['w'], b['w']) and torch.equal(a['gx'], b['gx'])


def test_vqkd_one_call_matches_the_reference_fixture_and_its_gradient():
    g = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    q = build(vqkd_cfg(K, D), w)
    assert q._one_call_step(torch.from_numpy(x).cuda()) is not None
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    z, loss, memo = q(xd, {})
    np.testing.assert_array_equal(memo['quant'].cpu().numpy(), g['quant'].astype(np.int64))
    np.testing.assert_allclose(q.embedding.weight.detach().cpu().numpy(), g['w_new'], rtol=0, atol=3e-6)
    torch.autograd.backward([loss, z], [None, gz])
    xt = torch.from_numpy(x).requires_grad_(True)
    xn = torch.nn.functional.normalize(xt)
    zt = tr.decode(torch.from_numpy(g['quant'].astype(np.int64)), torch.from_numpy(g['w_new']))
    rl = tr.commitment_loss(zt, xn, norm=True)
    zs = tr.ste(zt, xn)
    (rl + (zs * gz.cpu()).sum()).backward()
    assert abs(loss.item() - rl.item()) <= 1e-5
    np.testing.assert_allclose(z.detach().cpu().numpy(), zs.detach().numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(memo['x'].detach().cpu().numpy(), xn.detach().numpy(), rtol=0, atol=1e-7)


def test_forced_exchange_takes_the_two_phase_route(monkeypatch):
    """With an exchange to run and no communicator of the library's own, the forward is two calls around the caller's
    collective: forced here at one rank (the collective is then any callable; identity for a one-rank SUM)."""
    from vector_quantization_amd import train_step
    from vector_quantization_amd.quantizers import callbacks as cbm
    N, K, D = 2000, 1024, 64
    w0 = synth.unit_rows(synth.rng(9).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 4, 35)
    gz = torch.zeros(N, D, device='cuda')
    base = run_steps(build(cvq_cfg(K, D, 'Cosine'), w0), xs, gz, True)
    wk, xk, gk = w0[:, :32].copy(), [x[:, :32].contiguous() for x in xs], gz[:, :32].contiguous()
    base_k = run_steps(build(vqkd_cfg(K, 32), wk, True), xk, gk, True)
    calls = []
    monkeypatch.setattr(cbm, 'exchanging', lambda: True)
    import vector_quantization_amd.utils as U
    monkeypatch.setattr(U, 'all_reduce_sum', lambda t: calls.append(t.numel()) or t)
    forced = run_steps(build(cvq_cfg(K, D, 'Cosine'), w0), xs, gz, True)
    assert_same(base, forced)
    assert len(calls) == len(xs) and calls[0] == 2 * K + 4 + K * D and calls[-1] < calls[0]     # the payload shrinks with the list
    calls.clear()
    forced_k = run_steps(build(vqkd_cfg(K, 32), wk, True), xk, gk, True)
    assert len(calls) == len(xs) and calls[0] == 2 * K + 4 + K * 32
    assert_same(base_k, forced_k, exact_w=False)
    del train_step


def test_graphed_cvq_capacity_buckets_follow_the_list():
    """graphs.GraphedQuantizer captures the CVQ-VAE step at several capacities of the listed-code launches and chains the
    replays through the pinned count word: every step runs the smallest captured capacity that holds its list, and the
    results stay those of the eager module, through a change of bucket and through probabilities replaced from outside."""
    from vector_quantization_amd.graphs import GraphedQuantizer
    N, K, D = 3000, 2048, 64
    w0 = synth.unit_rows(synth.rng(5).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 8, 36)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) / (N * D)

    def steps(call, q, xs_):
        out = []
        for x in xs_:
            xin = x.clone().requires_grad_(True)
            q.zero_grad(set_to_none=True)
            z, loss, quant = call(xin)
            torch.autograd.backward([loss, z], [None, gz])
            out.append((quant.clone(), z.detach().clone(), float(loss), q.embedding.weight.detach().clone(),
                        q.get_buffer('_probability').clone()))
        return out

    q1 = build(cvq_cfg(K, D, 'Cosine'), w0)
    counts = []

    def eager(xin):
        z, loss, memo = q1(xin, {})
        counts.append(q1._callbacks.callbacks[0].last_exchange_rows)
        return z, loss, memo['quant']

    ref = steps(eager, q1, xs[:6])
    with torch.no_grad():                                         # probabilities replaced from outside, then two more steps
        q1.get_buffer('_probability').mul_(0.25)
    ref += steps(eager, q1, xs[6:])
    assert counts[0] == K and min(counts) < K
    caps = tuple(sorted({min(counts), sorted(counts)[len(counts) // 2]} - {K}))
    q2 = build(cvq_cfg(K, D, 'Cosine'), w0)
    gq = GraphedQuantizer(q2, xs[0], bucket_caps=caps)
    used = []

    def graphed(xin):
        out = gq(xin)
        used.append(gq.last_capacity)
        return out

    got = steps(graphed, q2, xs[:6])
    with torch.no_grad():
        q2.get_buffer('_probability').mul_(0.25)
    got += steps(graphed, q2, xs[6:])
    all_caps = list(caps) + [K]
    assert used == [min(c for c in all_caps if c >= n) for n in counts], (used, counts, caps)
    assert len(set(used)) >= 2, used
    for (qa, za, la, wa, pa), (qb, zb, lb, wb, pb) in zip(ref, got):
        assert torch.equal(qa, qb) and torch.equal(za, zb) and torch.equal(wa, wb) and torch.equal(pa, pb)
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la))


def test_graphed_cvq_replays_interleaved_with_eager_steps():
    """Round-5 advisor: an eager train step between two replays (a ragged last batch the graph refuses by shape) rewrites the
    device-side list in place without touching the early word or ``p._version`` — the next replay used to trust a stale
    length and could pick a capacity SMALLER than the list (anchors of the codes beyond it read from unwritten slots), and an
    eager step after a replay read a count the replay had not produced.  `CvqStepState.writer` names the last writer: both
    directions recount.  A second, smaller eager batch makes the list GROW between replays (codes fall out of use), which is
    exactly the case where the stale length is too small.  Compared with the all-eager module step by step."""
    from vector_quantization_amd.graphs import GraphedQuantizer
    N, K, D = 3000, 2048, 64
    w0 = synth.unit_rows(synth.rng(5).standard_normal((K, D), dtype=np.float32))
    xs = batches(N, K, D, w0, 10, 36)
    small = [x[:700].clone() for x in xs]                        # the "ragged last batch": another shape, eager only
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) / (N * D)
    plan = ['g', 'g', 'g', 'e', 'g', 'e', 'e', 'g', 'g', 'e']      # g: full batch (replayed in run B), e: small batch (eager in both)

    def run(graphed):
        q = build(cvq_cfg(K, D, 'Cosine'), w0)
        gq = None
        out, caps = [], []
        for i, kind in enumerate(plan):
            x = (xs[i] if kind == 'g' else small[i]).clone().requires_grad_(True)
            q.zero_grad(set_to_none=True)
            if kind == 'g' and graphed:
                if gq is None:
                    gq = GraphedQuantizer(q, xs[0], bucket_caps=(64, 256, 1024))
                z, loss, quant = gq(x)
                caps.append(gq.last_capacity)
            else:
                z, loss, memo = q(x, {})
                quant = memo['quant']
                caps.append(q._callbacks.callbacks[0].last_exchange_rows)
            torch.autograd.backward([loss, z], [None, gz[:x.shape[0]]])
            out.append((quant.clone(), z.detach().clone(), float(loss), q.embedding.weight.detach().clone(),
                        q.get_buffer('_probability').clone()))
        return out, caps

    ref, counts = run(False)
    got, used = run(True)
    for i, ((qa, za, la, wa, pa), (qb, zb, lb, wb, pb)) in enumerate(zip(ref, got)):
        assert torch.equal(qa, qb) and torch.equal(za, zb) and torch.equal(pa, pb), (i, plan[i])
        assert torch.equal(wa, wb), (i, plan[i], counts, used)
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la))
    for i, kind in enumerate(plan):                              # a replay never ran below the length of its list
        assert used[i] >= counts[i], (i, kind, counts, used)


@pytest.mark.parametrize('kind,dist,D,bf16,train', [('plain', 'L2', 256, True, True), ('plain', 'L2', 256, True, False), ('plain', 'Cosine', 32, False, True),
                                                    ('normalize', 'L2', 8, False, True), ('normalize', 'L2', 8, True, False),
                                                    ('normalize', 'Cosine', 64, False, True)])
def test_plain_and_normalize_forward_one_call_equals_hook_by_hook(kind, dist, D, bf16, train):
    """vqhip_vq_forward: a quantizer without an update callback (configs/vqgan/model.py:19-23) and one with NormalizeCallback
    alone (configs/llamagen/vqgan.py:18-20), train and eval — tokens, outputs, codebooks bit for bit, gradients to 1e-5."""
    N, K = 3000, 2048
    w0 = synth.rng(11).standard_normal((K, D), dtype=np.float32)
    xs = batches(N, K, D, synth.unit_rows(w0), 3, 37, bf16)
    gz = torch.randn(N, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(6)) / (N * D)
    cfg = dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D), distance=dict(type=f'{dist}Distance'),
               losses=dict(vqgan_loss=dict(type='VQGANLoss')), callbacks=[dict(type='NormalizeCallback')] if kind == 'normalize' else [])
    recs = []
    for one_call in (False, True):
        q = build(cfg, w0)
        q.train(train)
        assert (q._one_call_step(xs[0]) is not None) == one_call or not one_call
        q.one_call_steps = one_call
        assert (q._one_call_step(xs[0]) is not None) == one_call
        rec = []
        for x in xs:
            xin = x.clone().requires_grad_(train)
            for p in q.parameters():
                p.grad = None
            z, loss, memo = q(xin, {})
            if train:
                torch.autograd.backward([loss, z], [None, gz])
            wg = q.embedding.weight.grad
            rec.append(dict(quant=memo['quant'].clone(), z=z.detach().clone(), loss=loss.detach().clone(),
                            gx=xin.grad.clone() if train else torch.zeros(1), gw=None if wg is None else wg.clone(),
                            w=q.embedding.weight.detach().clone(), hist=memo['encode'].get('hist', torch.zeros(1)).clone(),
                            memo_x=memo['x'].detach().clone(), loss_memo={k: v.detach().clone() for k, v in memo['loss'].items()}))
            assert memo['encode']['distance'].shape == (N, K)
        recs.append(rec)
    assert_same(recs[0], recs[1])


@pytest.mark.parametrize('metric,D,dtype', [('Cosine', 256, torch.float32), ('L2', 256, torch.float32), ('CosineBF16', 64, torch.float32),
                                            ('L2', 8, torch.float32), ('Cosine', 768, torch.float32), ('L2', 64, torch.bfloat16),
                                            ('Cosine', 64, torch.bfloat16)])
def test_column_pass_over_a_short_list_direct_form_equals_the_pipeline(metric, D, dtype):
    """vqhip_col_argmin_rows on a short list runs the definition's own fp32 pass over the listed codes instead of the
    role-swapped proposal pipeline (tuning key 15): identical indices, also against a brute-force float64 column argmin;
    bf16 latents keep the pipeline under either metric (the pass reads fp32 rows: round-5 advisor — the cosine case read the
    bf16 buffer as fp32)."""
    from vector_quantization_amd import _lib, ops
    N, K = 3072, 4096
    g = torch.Generator(device='cuda').manual_seed(D)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = (w[torch.randint(0, K, (N,), device='cuda', generator=g)] + 0.3 * torch.randn(N, D, device='cuda', generator=g)).to(dtype)
    p = torch.full((K,), 1.0 / K, device='cuda')
    listed = torch.randperm(K, device='cuda', generator=g)[:57].sort().values
    p[listed] = 0.0
    rows, slot, count = ops.cvq_rows(p, K, 0.99, 1e-3)
    assert int(count) == 57 and torch.equal(rows[:57].long(), listed)
    if metric.startswith('Cosine'):
        xq, eq = ops.normalize_rows(x), ops.normalize_rows(w)
        if metric == 'CosineBF16':
            xq, eq = xq.bfloat16().float(), eq.bfloat16().float()
        elif dtype == torch.bfloat16:
            xq = xq.bfloat16()                                   # bf16 rows handed to the cosine column pass as they are
    else:
        xq, eq = x, w
    L = _lib.lib()
    outs = []
    for direct in (1, 0):
        L.vqhip_set_tuning(15, direct)
        outs.append(ops.col_argmin_rows(xq, eq, rows, count, 64, metric)[:57].clone())
    L.vqhip_set_tuning(15, 1)
    assert torch.equal(outs[0], outs[1])
    if metric != 'CosineBF16':
        xe, ee = xq.double(), eq[listed].double()
        d = torch.cdist(ee, xe) if metric == 'L2' else 1 - ee @ xe.t()
        best = d.min(1).values
        got = d.gather(1, outs[0].reshape(-1, 1)).reshape(-1)
        assert bool(((got - best).abs() <= 1e-5 * best.abs().clamp_min(1e-3)).all())      # the definition's argmin up to fp32 near-ties
