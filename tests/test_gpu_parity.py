"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden fixtures.

Bar: token indices bit-exact against the oracle on every case (the oracle fixes the fp32 summation
order, see oracle/vq_oracle.c) and against the reference-op fixtures wherever those are well conditioned;
gather/STE bit-exact; losses within 1e-5 (fp32), the tolerance BASELINE.json's north_star states.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
EXACT_KINDS = {'int', 'normal', 'planted', 'unit', 'normal_bf16x'}


def _encode_files():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, '*.npz'))):
        z = np.load(p)
        if 'spec' in z.files and 'quant' in z.files and 'kind' in json.loads(str(z['spec'])) \
                and json.loads(str(z['spec'])).get('distance') != 'CosineBF16':
            out.append(p)
    return out


AUTOCAST_FILES = sorted(glob.glob(os.path.join(GOLDEN, 'cosbf16_*.npz')))


ENCODE_FILES = _encode_files()


@pytest.fixture(scope='module')
def ops():
    from vector_quantization_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def gpu_encode(ops, spec, x, w, x_dtype=None):
    """Mirror of the quantizer's encode: optional NormalizeCallback, distance-specific preparation, argmin."""
    xd, wd = dev(x, x_dtype), dev(w)
    if spec['normalize']:
        xd, wd = ops.normalize_rows(xd), ops.normalize_rows(wd)
    metric = spec['distance']
    xq = ops.normalize_rows(xd) if metric == 'Cosine' else xd
    cb = ops.prepare_codebook(wd, metric)
    idx, st = ops.argmin(xq, cb, return_stats=True)
    return xd, wd, xq, cb, idx, st


@pytest.mark.parametrize('path', ENCODE_FILES, ids=[os.path.basename(p)[:-4] for p in ENCODE_FILES])
def test_encode_matches_oracle_and_golden(ops, path):
    z = np.load(path)
    spec = json.loads(str(z['spec']))
    x, w = synth.make_inputs(spec['kind'], spec['seed'], spec['N'], spec['K'], spec['D'])
    assert synth.sha(x) == str(z['x_sha'])
    xo, wo = (co.normalize_rows(x), co.normalize_rows(w)) if spec['normalize'] else (x, w)
    oracle_idx = (co.l2_argmin if spec['distance'] == 'L2' else co.cos_argmin)(xo, wo)

    xd, wd, xq, cb, idx, st = gpu_encode(ops, spec, x, w)
    torch.cuda.synchronize()
    if spec['normalize']:       # the normalisation kernel itself is bit-exact against the oracle's order
        np.testing.assert_array_equal(xd.cpu().numpy(), xo)
        np.testing.assert_array_equal(wd.cpu().numpy(), wo)
    got = idx.cpu().numpy()
    np.testing.assert_array_equal(got, oracle_idx)                       # bit-exact vs oracle, every case
    tiny = spec['N'] <= 25 and spec['K'] <= 25
    # and vs the reference's own ATen ops (fixture): every row of every case, except 8 of the 512 rows of the ill-conditioned
    # U(+-1/K) init at K = 16 384 (two GEMM orders differ there by <= 1 ulp of distance: tests/test_oracle_golden.py)
    known = {'l2_c2_vqganinit_s3407': 8}
    assert int((got != z['quant'].astype(np.int64)).sum()) == known.get(spec['name'], 0)
    # the fp32-only entry point agrees bit for bit
    wq = ops.normalize_rows(wd) if spec['distance'] == 'Cosine' else wd
    idx2, dmin = ops.argmin_exact(xq, wq, spec['distance'], return_min=True)
    np.testing.assert_array_equal(idx2.cpu().numpy(), oracle_idx)
    o_idx, o_min = (co.l2_argmin if spec['distance'] == 'L2' else co.cos_argmin)(xo, wo, with_min=True)
    np.testing.assert_array_equal(dmin.cpu().numpy(), o_min)
    # histogram fused into the epilogue
    hist = torch.zeros(spec['K'], dtype=torch.int32, device='cuda')
    ops.argmin(xq, cb, hist=hist)
    np.testing.assert_array_equal(hist.cpu().numpy().astype(np.int64), co.bincount(oracle_idx, spec['K']))
    # decode + STE bit-exact, loss within 1e-5
    zt, zs, sse = ops.gather_ste_loss(xd, wd, idx)
    zo, zso = co.gather_ste(xo, wo, oracle_idx)
    np.testing.assert_array_equal(zt.cpu().numpy(), zo)
    np.testing.assert_array_equal(zs.cpu().numpy(), zso)
    mse = float(sse.item()) / (spec['N'] * spec['D'])
    ref = float(co.mse(zo, xo))
    assert abs(mse - ref) <= 1e-5 * max(1.0, abs(ref))
    # the same pass with the mean finished on the device: identical value, twice (the scratch comes back zeroed)
    want = (sse / (spec['N'] * spec['D'])).float()
    for _ in range(2):
        _, zs2, m2 = ops.gather_ste_mse(xd, wd, idx, beta=0.25)
        assert torch.equal(zs2, zs) and m2.dtype == torch.float32
        assert float(m2[0]) == float(want) and float(m2[1]) == float(want)
        assert float(m2[2]) == float(want + want * 0.25)            # VQGANLoss: codebook + beta * commitment, two roundings
    if spec['loss'] == 'vqgan' and not spec['normalize'] and spec['kind'] in EXACT_KINDS and not tiny:
        assert abs(1.25 * mse - float(z['loss'])) <= 1e-5 * max(1.0, abs(float(z['loss'])))
    print(f"{spec['name']}: rescan={int(st[0])} multi={int(st[1])} exact={int(st[2])} of {spec['N']}")


def test_bf16_latents_same_as_fp32_values(ops):
    z = np.load(os.path.join(GOLDEN, 'l2_c2_bf16x_s3407.npz'))
    spec = json.loads(str(z['spec']))
    x, w = synth.make_inputs(spec['kind'], spec['seed'], spec['N'], spec['K'], spec['D'])
    _, wd, _, cb, idx, _ = gpu_encode(ops, spec, x, w, x_dtype=torch.bfloat16)   # x is exactly bf16-representable
    np.testing.assert_array_equal(idx.cpu().numpy(), z['quant'].astype(np.int64))
    zt, zs, sse = ops.gather_ste_loss(dev(x, torch.bfloat16), wd, idx)
    zo, zso = co.gather_ste(x, w, z['quant'].astype(np.int64))
    np.testing.assert_array_equal(zs.cpu().numpy(), zso)


def test_nonfinite_inputs(ops):
    z = np.load(os.path.join(GOLDEN, 'special_nonfinite.npz'))
    x, w, wn = z['x'], z['w'], z['w_nan']
    for ww, key in ((w, 'quant_l2'), (wn, 'quant_l2_wnan')):
        cb = ops.prepare_codebook(dev(ww), 'L2')
        idx = ops.argmin(dev(x), cb)
        np.testing.assert_array_equal(idx.cpu().numpy(), z[key].astype(np.int64))
        np.testing.assert_array_equal(ops.argmin_exact(dev(x), dev(ww), 'L2').cpu().numpy(), z[key].astype(np.int64))
    xq = ops.normalize_rows(dev(x))
    cb = ops.prepare_codebook(dev(w), 'Cosine')
    np.testing.assert_array_equal(ops.argmin(xq, cb).cpu().numpy(), z['quant_cos'].astype(np.int64))


def test_sqrt_buckets_tie_to_the_lowest_index(ops):
    """Distinct radicands whose square roots round to the same float tie (torch.cdist takes the root before argmin compares),
    and the winner is the LOWEST index even if its radicand is the larger one.  The all-fp32 route compares radicands first
    (vqhip_exact_kernels.h): this is the case it must hand to its exact loop.  Rows of zeros make the radicand |e_k|^2:
    code `hi` = (1.25, 0, ...) has 1.5625, code `lo` = (1.25, 2^-11.5, 0, ...) the next float above it; both roots are 1.25."""
    K, D, lo, hi = 300, 8, 17, 211
    rng = np.random.default_rng(5)
    w = (3.0 + rng.random((K, D))).astype(np.float32)                        # every other code is far away
    w[lo] = 0; w[hi] = 0
    w[hi, 0] = 1.25
    w[lo, 0] = 1.25; w[lo, 1] = np.float32(2.0 ** -11.5)
    x = np.zeros((5, D), np.float32)
    d = co.l2_dist(x[:1], w)[0]
    en = (w[[lo, hi]].astype(np.float64) ** 2).sum(1)
    assert en[0] > en[1] and d[lo] == d[hi] == np.float32(1.25)              # larger radicand, same float distance
    ref = co.l2_argmin(x, w)
    assert (ref == lo).all()
    np.testing.assert_array_equal(ops.argmin_exact(dev(x), dev(w), 'L2').cpu().numpy(), ref)
    np.testing.assert_array_equal(ops.argmin(dev(x), ops.prepare_codebook(dev(w), 'L2')).cpu().numpy(), ref)
    # and with the roles of the indices swapped the smaller radicand is also the lower index
    w[[lo, hi]] = w[[hi, lo]]
    np.testing.assert_array_equal(ops.argmin_exact(dev(x), dev(w), 'L2').cpu().numpy(), co.l2_argmin(x, w))


@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
def test_distance_matrix_and_col_argmin(ops, metric):
    x, w = synth.make_inputs('normal', 77, 200, 333, 32)
    if metric == 'Cosine':
        xo, wo = co.normalize_rows(x), co.normalize_rows(w)
        d_ref = co.cos_dist(x, w)
    else:
        xo, wo = x, w
        d_ref = co.l2_dist(x, w)
    d = ops.distance(dev(xo), dev(wo), metric)
    np.testing.assert_array_equal(d.cpu().numpy(), d_ref)
    col = ops.col_argmin(dev(xo), dev(wo), metric)
    np.testing.assert_array_equal(col.cpu().numpy(), co.col_argmin(d_ref))


def test_row_kernels(ops):
    v = synth.normal(5, 1000, 256)
    np.testing.assert_array_equal(ops.row_sqnorm(dev(v)).cpu().numpy(), co.row_sqnorm(v))
    for D in (8, 32, 48, 256, 768):
        v = synth.normal(6, 257, D)
        v[3] = 0
        np.testing.assert_array_equal(ops.normalize_rows(dev(v)).cpu().numpy(), co.normalize_rows(v))


def test_prepared_cosine_image_holds_the_normalised_rows(ops):
    """PreparedCodebook.exact_rows(): the view NearestAnchor's column pass reads instead of normalising again."""
    for K, D in ((1000, 32), (4096, 256), (37, 24)):
        w = torch.randn(K, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(K + D)) * 3.0
        cb = ops.prepare_codebook(w, 'Cosine')
        assert torch.equal(cb.exact_rows(), ops.normalize_rows(w))
        assert ops.prepare_codebook(w, 'L2').exact_rows() is None


@pytest.mark.parametrize('N,K,D,metric,dtype', [(3072, 16384, 256, 'Cosine', None), (8192, 16384, 256, 'L2', torch.bfloat16),
                                                (20000, 8192, 32, 'Cosine', None), (1000, 777, 24, 'L2', None),
                                                (50000, 4099, 8, 'L2', None), (513, 100, 768, 'Cosine', torch.bfloat16),
                                                (300, 50, 1032, 'L2', None), (31, 5, 520, 'Cosine', None)])
def test_encode_one_call_equals_separate_calls(ops, N, K, D, metric, dtype):
    """vqhip_encode (codebook statistics + token side in one launch, cosine normalisation folded in) against
    prepare_codebook / normalize_rows / argmin: indices, histogram, normalised rows and the prepared image's fp32 rows."""
    g = torch.Generator(device='cuda').manual_seed(N * 7 + K)
    w = torch.randn(K, D, device='cuda', generator=g) * 0.7
    x = torch.randn(N, D, device='cuda', generator=g) * 3.0
    x[::5] = w[torch.randint(0, K, (len(x[::5]),), device='cuda', generator=g)] * 1.5
    x[1] = 0
    if dtype is not None:
        x = x.to(dtype)
    cb = ops.prepare_codebook(w, metric)
    xq_ref = ops.normalize_rows(x) if metric == 'Cosine' else x
    h_ref = torch.zeros(K, dtype=torch.int32, device='cuda')
    ref = ops.argmin(xq_ref, cb, hist=h_ref)
    h = torch.zeros(K, dtype=torch.int32, device='cuda')
    got, cb2, xq = ops.encode(x, w, metric, hist=h)
    assert torch.equal(got, ref) and torch.equal(h, h_ref)
    if metric == 'Cosine':
        assert torch.equal(xq, xq_ref) and torch.equal(cb2.exact_rows(), cb.exact_rows())
    else:
        assert xq is None
    # zero_hist: the histogram buffer may hold anything, the call's first launch zeroes it
    hg = torch.full((K,), 12345, dtype=torch.int32, device='cuda')
    got2, _, _ = ops.encode(x, w, metric, hist=hg, zero_hist=True)
    assert torch.equal(got2, ref) and torch.equal(hg, h_ref)
    # the image made by the fused front serves later argmin calls like any other
    assert torch.equal(ops.argmin(xq_ref, cb2), ref)
    wq = ops.normalize_rows(w) if metric == 'Cosine' else w
    xe = xq_ref.float() if metric == 'Cosine' else x
    assert torch.equal(got, ops.argmin_exact(xe, wq, metric))


@pytest.mark.parametrize('path', AUTOCAST_FILES, ids=[os.path.basename(p)[:-4] for p in AUTOCAST_FILES])
def test_bf16_autocast_cosine_matches_reference(ops, path):
    """VQHIP_METRIC_COS_BF16 (opt-in: CosineDistance(autocast='bf16')) against the reference's own CosineDistance run
    inside torch.autocast(bf16) (fixtures) and against the oracle's definition: row argmin through the one-call
    encode, the separate calls and the fp32-only route, the bf16-valued minima, and NearestAnchor's column argmin."""
    z = np.load(path)
    spec = json.loads(str(z['spec']))
    x, w = synth.make_inputs(spec['kind'], spec['seed'], spec['N'], spec['K'], spec['D'])
    want = z['quant'].astype(np.int64)
    xd, wd = dev(x), dev(w)
    got, cb, xq = ops.encode(xd, wd, 'CosineBF16')
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    xn = ops.normalize_rows(xd).bfloat16().float()
    wn = ops.normalize_rows(wd).bfloat16().float()
    assert torch.equal(xq, xn) and torch.equal(cb.exact_rows(), wn)
    np.testing.assert_array_equal(ops.argmin(xn, ops.prepare_codebook(wd, 'CosineBF16')).cpu().numpy(), want)
    idx2, dmin = ops.argmin_exact(xn, wn, 'CosineBF16', return_min=True)
    np.testing.assert_array_equal(idx2.cpu().numpy(), want)
    np.testing.assert_array_equal(dmin.cpu().numpy(), z['mind'])
    np.testing.assert_array_equal(ops.col_argmin(xn, wn, 'CosineBF16').cpu().numpy(), z['col_idx'].astype(np.int64))
    o_idx, o_min = co.cos_bf16_argmin(x, w, with_min=True)
    np.testing.assert_array_equal(o_idx, want)
    # the materialised matrix holds the bf16 values of the reference's distance tensor
    d = ops.distance(xn, wn, 'CosineBF16')
    assert torch.equal(d, d.bfloat16().float())
    np.testing.assert_array_equal(d.min(1).values.cpu().numpy(), z['mind'])


def test_bf16_autocast_cosine_module_and_larger_shapes(ops):
    """The quantizer module with CosineDistance(autocast='bf16') on BASELINE configs[2]'s shape (and a D = 256 one) against
    the oracle's definition on a row sample and against the fp32-only route everywhere."""
    from vector_quantization_amd import build_quantizer, Config
    for N, K, D in ((100352, 8192, 32), (20000, 16384, 256)):
        g = torch.Generator(device='cuda').manual_seed(N + D)
        w = torch.randn(K, D, device='cuda', generator=g)
        x = torch.randn(N, D, device='cuda', generator=g)
        q = build_quantizer(dict(type='VQGANQuantizer',
                                 embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                                 distance=dict(type='CosineDistance', autocast='bf16'),
                                 losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
        q.init_weights(Config(type='vqgan')); q = q.cuda().eval()
        with torch.no_grad():
            q.embedding.weight.copy_(w)
            _, _, memo = q(x, {})
        quant = memo['quant'].reshape(-1)
        xn = ops.normalize_rows(x).bfloat16().float()
        wn = ops.normalize_rows(w).bfloat16().float()
        assert torch.equal(quant, ops.argmin_exact(xn, wn, 'CosineBF16'))
        sample = slice(0, 400)
        np.testing.assert_array_equal(quant[sample].cpu().numpy(), co.cos_bf16_argmin(x[sample].cpu().numpy(), w.cpu().numpy()))
        fp32 = ops.encode(x, w, 'Cosine')[0]
        assert 0 < int((fp32 != quant).sum()) < N // 5        # the mode changes a few percent of the rows, not most


def test_hist_scatter_gather(ops):
    g = synth.rng(9)
    N, K, D = 5000, 300, 32
    idx = g.integers(0, K, N)
    src = synth.normal(10, N, D)
    np.testing.assert_array_equal(ops.hist(dev(idx), K).cpu().numpy().astype(np.int64), co.bincount(idx, K))
    out = ops.scatter_add_rows(dev(src), dev(idx), K).cpu().numpy()
    np.testing.assert_allclose(out, co.scatter_add_rows(src, idx, K), rtol=1e-5, atol=1e-4)
    rows = g.integers(0, N, K)
    np.testing.assert_array_equal(ops.gather_rows(dev(src), dev(rows)).cpu().numpy(), src[rows])


def test_vqkd_update_matches_golden(ops):
    z = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    spec = json.loads(str(z['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    xn = ops.normalize_rows(dev(x))
    cb = ops.prepare_codebook(dev(w), 'Cosine')
    quant = ops.argmin(ops.normalize_rows(xn), cb)
    np.testing.assert_array_equal(quant.cpu().numpy(), z['quant'].astype(np.int64))
    xs = ops.normalize_rows(xn)                                   # callbacks.py:124
    hist = ops.hist(quant, K).long()
    sums = ops.scatter_add_rows(xs, quant, K)
    wd = dev(w).clone()
    ops.vqkd_update_(wd, hist, sums, 0.99)
    np.testing.assert_allclose(wd.cpu().numpy(), z['w_new'], rtol=0, atol=3e-6)


@pytest.mark.parametrize('dist', ['L2', 'Cosine'])
def test_cvq_update_matches_golden(ops, dist):
    z = np.load(os.path.join(GOLDEN, f'update_cvq_{dist.lower()}.npz'))
    spec = json.loads(str(z['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    xd, wd = dev(x), dev(w).clone()
    p = torch.zeros(K, device='cuda')
    for step, (qk, ck, pk, wk) in enumerate((('quant', 'col_idx', 'p1', 'w_new'), ('quant2', 'col_idx2', 'p2', 'w_new2'))):
        if dist == 'Cosine':
            xq, wq = ops.normalize_rows(xd), ops.normalize_rows(wd)
        else:
            xq, wq = xd, wd
        cb = ops.prepare_codebook(wd, dist)
        quant = ops.argmin(xq, cb)
        col = ops.col_argmin(xq, wq, dist)
        if step == 0:      # identical inputs: indices are bit-exact; step 2 starts from a tolerance-equal codebook
            np.testing.assert_array_equal(quant.cpu().numpy(), z[qk].astype(np.int64))
            np.testing.assert_array_equal(col.cpu().numpy(), z[ck].astype(np.int64))
        hist = ops.hist(quant, K).long()
        anchors = ops.gather_rows(xd, col)
        # the one-launch form (vqhip_cvq_step) on the int32 histogram: bit-identical to the staged update, out of place
        # and in place, fp32 and bf16 latents
        w1, p1 = torch.empty_like(wd), torch.empty_like(p)
        ops.cvq_step(wd, w1, p, p1, hist.int(), N, xd, col, 0.99, 1e-3)
        w2, p2 = wd.clone(), p.clone()
        ops.cvq_step(w2, w2, p2, p2, hist.int(), N, xd, col, 0.99, 1e-3)
        xb = xd.bfloat16()
        wb_ref, pb_ref = wd.clone(), p.clone()
        ops.cvq_update_(wb_ref, pb_ref, hist, N, ops.gather_rows(xb, col), 0.99, 1e-3)
        wb, pb = torch.empty_like(wd), torch.empty_like(p)
        ops.cvq_step(wd, wb, p, pb, hist.int(), N, xb, col, 0.99, 1e-3)
        ops.cvq_update_(wd, p, hist, N, anchors, 0.99, 1e-3)
        assert torch.equal(w1, wd) and torch.equal(p1, p) and torch.equal(w2, wd) and torch.equal(p2, p)
        assert torch.equal(wb, wb_ref) and torch.equal(pb, pb_ref)
        np.testing.assert_allclose(p.cpu().numpy(), z[pk], rtol=1e-5, atol=1e-8)
        if step == 0:
            np.testing.assert_allclose(wd.cpu().numpy(), z[wk], rtol=0, atol=3e-6)


# ---- BASELINE.json sizes: size-independent properties --------------------------------------------------

def test_full_size_properties(ops):
    """K=16384, D=256, N=65536 (configs[1]/[4] shapes): decode->encode round trip, proposal pass == fp32 pass,
    histogram mass, and a checksum against the oracle on a row sample."""
    K, D, N = 16384, 256, 65536
    g = torch.Generator(device='cuda').manual_seed(3407)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
    cb = ops.prepare_codebook(w, 'L2')
    hist = torch.zeros(K, dtype=torch.int32, device='cuda')
    idx, st = ops.argmin(x, cb, hist=hist, return_stats=True)
    assert int(hist.sum()) == N and int(idx.min()) >= 0 and int(idx.max()) < K
    # (1) against the fp32-only pass on the whole batch
    idx_exact = ops.argmin_exact(x, w, 'L2')
    assert torch.equal(idx, idx_exact)
    # (2) round trip: encoding the decoded codes returns the codes (codebook rows are distinct)
    perm = torch.randperm(K, device='cuda', generator=g)
    z, _, _ = ops.gather_ste_loss(torch.zeros(K, D, device='cuda'), w, perm, need_ste=False, need_sse=False)
    assert torch.equal(ops.argmin(z, cb), perm)
    # (3) idempotence: quantising the quantised latents changes nothing
    zq, _, _ = ops.gather_ste_loss(x, w, idx, need_ste=False, need_sse=False)
    assert torch.equal(ops.argmin(zq, cb), idx)
    # (4) oracle on a sample of rows
    rows = torch.arange(0, N, 257, device='cuda')
    xs = x[rows].float().cpu().numpy()
    np.testing.assert_array_equal(idx[rows].cpu().numpy(), co.l2_argmin(xs, w.cpu().numpy()))
    print(f'full size: rescan={int(st[0])} multi={int(st[1])} exact={int(st[2])} of {N}')


def test_empty_and_ragged(ops):
    w = dev(synth.normal(1, 100, 16))
    cb = ops.prepare_codebook(w, 'L2')
    assert ops.argmin(torch.empty(0, 16, device='cuda'), cb).numel() == 0
    for N in (1, 31, 33, 257):
        x = synth.normal(N, N, 16)
        np.testing.assert_array_equal(ops.argmin(dev(x), cb).cpu().numpy(), co.l2_argmin(x, w.cpu().numpy()))


@pytest.mark.parametrize('D', [6, 8, 12, 16, 24, 30, 64, 128, 254, 512, 520, 640, 768, 776, 1024, 1030, 1032])
@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
def test_all_supported_dims(ops, D, metric):
    """Every padded-D instantiation of the proposal kernel (k-steps of 16: 2..32 pipelined, 48 and 64 with one tile per
    stage) and the fp32-only route (D % 8 != 0 or D > 1024) in both of its forms (D % 4 == 0: whole 16-byte pieces; any D:
    element-wise tails); the all-fp32 route itself on every D as well."""
    N, K = 700, 1500
    x, w = synth.make_inputs('normal', 100 + D, N, K, D)
    if metric == 'Cosine':
        ref = co.cos_argmin(x, w)
        xq = ops.normalize_rows(dev(x))
    else:
        ref = co.l2_argmin(x, w)
        xq = dev(x)
    cb = ops.prepare_codebook(dev(w), metric)
    np.testing.assert_array_equal(ops.argmin(xq, cb).cpu().numpy(), ref)
    wq = ops.normalize_rows(dev(w)) if metric == 'Cosine' else dev(w)
    np.testing.assert_array_equal(ops.argmin_exact(xq, wq, metric).cpu().numpy(), ref)


@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
def test_d1024_one_wave_per_simd_form(ops, metric):
    """D = 1024 has two proposal kernels, chosen per call by whole rounds of workgroups (vqhip.hip, `case 64`): 24 576 rows
    against 1024 codes is a shape that takes the four-wave form (one wave per SIMD, 48 tokens in registers, fragments pinned
    half to accumulation and half to vector registers); `test_all_supported_dims` (700 rows) takes the eight-wave form.
    Every row against the all-fp32 route, a sample against the C oracle."""
    N, K, D = 24576, 1024, 1024
    g = torch.Generator(device='cuda').manual_seed(1024)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g)
    x[::97] = w[torch.randint(0, K, (len(range(0, N, 97)),), device='cuda', generator=g)] + 1e-3 * torch.randn(len(range(0, N, 97)), D, device='cuda', generator=g)
    if metric == 'Cosine':
        xq, wq = ops.normalize_rows(x), ops.normalize_rows(w)
    else:
        xq, wq = x, w
    idx = ops.argmin(xq, ops.prepare_codebook(w, metric))
    assert torch.equal(idx, ops.argmin_exact(xq, wq, metric))
    rows = torch.arange(0, N, 389, device='cuda')
    xs, wn = x[rows].cpu().numpy(), w.cpu().numpy()
    ref = co.cos_argmin(xs, wn) if metric == 'Cosine' else co.l2_argmin(xs, wn)
    np.testing.assert_array_equal(idx[rows].cpu().numpy(), ref)


def test_large_batch_many_slices_and_single_slice(ops):
    """Slice counts 1..16 give identical indices (tuning knob 2 forces the split)."""
    from vector_quantization_amd import _lib
    L = _lib.lib()
    N, K, D = 3000, 4096, 64
    x, w = synth.make_inputs('normal', 55, N, K, D)
    ref = co.l2_argmin(x, w)
    cb = ops.prepare_codebook(dev(w), 'L2')
    try:
        for ns in (1, 2, 4, 8, 16):
            L.vqhip_set_tuning(2, ns)
            for pipe in (0, 1):
                L.vqhip_set_tuning(0, pipe)
                np.testing.assert_array_equal(ops.argmin(dev(x), cb).cpu().numpy(), ref)
    finally:
        L.vqhip_set_tuning(2, 0)
        L.vqhip_set_tuning(0, 1)


@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
@pytest.mark.parametrize('dtype', [None, torch.bfloat16])
def test_col_argmin_fast_path(ops, metric, dtype):
    """NearestAnchor at a CVQ-like shape: the role-swapped proposal pipeline equals the oracle's d.argmin(0),
    including duplicated latents (lowest token index wins) and codes nobody is near."""
    N, K, D = 3000, 4096, 256
    x, w = synth.make_inputs('normal', 91, N, K, D)
    x[1500] = x[7]                       # duplicated latents: ties between token 7 and 1500
    x[2999] = x[7]
    w[5] = x[7]                          # a code sitting exactly on them (distance 0)
    if dtype is not None:
        x = synth.bf16_round(x)
        w[5] = x[7]
    if metric == 'Cosine':
        xo, wo = co.normalize_rows(x), co.normalize_rows(w)
        d_ref = co.cos_dist(x, w)
        col = ops.col_argmin(dev(xo), dev(wo), metric)
    else:
        d_ref = co.l2_dist(x, w)
        col = ops.col_argmin(dev(x, dtype), dev(w), metric)
    ref = co.col_argmin(d_ref)
    np.testing.assert_array_equal(col.cpu().numpy(), ref)
    assert ref[5] == 7


@pytest.mark.parametrize('N,K,D,metric', [(2500, 20000, 32, 'Cosine'), (3000, 18000, 8, 'L2'), (1200, 16500, 16, 'Cosine')])
def test_col_argmin_small_d_many_codes(ops, N, K, D, metric):
    """NearestAnchor's role-swapped pass with >= 16 384 codes as its "rows" and D <= 32: the group-record proposal kernel with
    identify32_kernel behind it (dot-product metric, no aux reads) — against the materialised fp32 matrix's d.argmin(0)."""
    g = torch.Generator(device='cuda').manual_seed(N + K + D)
    x = torch.randn(N, D, device='cuda', generator=g)
    w = torch.randn(K, D, device='cuda', generator=g)
    w[17] = x[3]; x[N - 1] = x[3]                 # a code on a duplicated latent: the lower token index wins
    if metric == 'Cosine':
        x, w = ops.normalize_rows(x), ops.normalize_rows(w)
    col = ops.col_argmin(x, w, metric)
    d = ops.distance(x, w, metric).cpu().numpy()
    np.testing.assert_array_equal(col.cpu().numpy(), d.argmin(0))          # numpy: first occurrence on ties
    assert int(col[17]) == 3


@pytest.mark.parametrize('name,N,K,D,metric,normalize', [
    ('C3 VQ-KD', 512 * 196, 8192, 32, 'Cosine', False),          # configs[2]: K=8192 D=32, 14x14 tokens, batch 512
    ('C5 LlamaGen', 65536, 16384, 8, 'L2', True),                 # configs[4] reference shape: D=8 + NormalizeCallback
])
def test_full_size_other_configs(ops, name, N, K, D, metric, normalize):
    """BASELINE.json configs[2] / configs[4] at full size: the proposal pipeline equals the fp32-only pass on every
    row, the histogram has the right mass, and a row sample matches the CPU oracle."""
    g = torch.Generator(device='cuda').manual_seed(3407)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g)
    if normalize:
        x, w = ops.normalize_rows(x), ops.normalize_rows(w)
    xq = ops.normalize_rows(x) if metric == 'Cosine' else x
    wq = ops.normalize_rows(w) if metric == 'Cosine' else w
    cb = ops.prepare_codebook(w, metric)
    hist = torch.zeros(K, dtype=torch.int32, device='cuda')
    idx, st = ops.argmin(xq, cb, hist=hist, return_stats=True)
    assert int(hist.sum()) == N
    assert torch.equal(idx, ops.argmin_exact(xq, wq, metric))
    rows = torch.arange(0, N, 997, device='cuda')
    fn = co.cos_argmin if metric == 'Cosine' else co.l2_argmin
    np.testing.assert_array_equal(idx[rows].cpu().numpy(), fn(x[rows].cpu().numpy(), w.cpu().numpy()))
    print(f'{name}: rescan={int(st[0])} multi={int(st[1])} exact={int(st[2])} of {N}')


def test_hip_graph_capture_and_replay(ops):
    """The whole step (prepare + argmin + gather/STE/loss) is capture-safe: no allocation, sync or cross-stream work
    inside the library; a captured graph replays to the same indices on new data in the same buffers."""
    N, K, D = 4096, 2048, 64
    x, w = synth.make_inputs('normal', 71, N, K, D)
    x2 = synth.normal(72, N, D)
    xd, wd = dev(x), dev(w)

    def step():
        cb = ops.prepare_codebook(wd, 'L2')
        idx = ops.argmin(xd, cb)
        _, zs, sse = ops.gather_ste_loss(xd, wd, idx, need_z=False)
        return idx, zs, sse

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        idx, zs, sse = step()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx.cpu().numpy(), co.l2_argmin(x, w))
    xd.copy_(torch.from_numpy(x2))          # new batch in the captured input buffer
    graph.replay()
    torch.cuda.synchronize()
    ref2 = co.l2_argmin(x2, w)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref2)
    np.testing.assert_array_equal(zs.cpu().numpy(), co.gather_ste(x2, w, ref2)[1])


def test_mass_duplicates_overflow_to_fp32_pass(ops):
    """Hundreds of identical codes (dead codes collapsed on one point, as after a bad k-means init): every duplicate is a
    candidate, the per-row candidate list overflows, and the last-resort fp32 pass must return the LOWEST duplicate."""
    N, K, D = 512, 2048, 64
    x, w = synth.make_inputs('normal', 13, N, K, D)
    dup = np.arange(300, 300 + 400)
    w[dup] = w[300]                                  # 400 copies of code 300
    x[:64] = w[300] + 0.01 * x[:64]                  # 64 tokens right next to it
    ref = co.l2_argmin(x, w)
    assert (ref[:64] == 300).all()
    cb = ops.prepare_codebook(dev(w), 'L2')
    idx, st = ops.argmin(dev(x), cb, return_stats=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    assert int(st[2]) >= 64, f'expected the duplicated rows in the fp32 pass, stats={st.tolist()}'
    # cosine + histogram through the same path
    refc = co.cos_argmin(x, w)
    cbc = ops.prepare_codebook(dev(w), 'Cosine')
    hist = torch.zeros(K, dtype=torch.int32, device='cuda')
    idxc = ops.argmin(ops.normalize_rows(dev(x)), cbc, hist=hist)
    np.testing.assert_array_equal(idxc.cpu().numpy(), refc)
    np.testing.assert_array_equal(hist.cpu().numpy().astype(np.int64), co.bincount(refc, K))


def test_randomised_shapes_sweep(ops):
    """Random (N, K, D, metric, dtype, distribution) draws against the oracle: ragged tiles, tiny codebooks, duplicated
    rows, scaled data."""
    g = synth.rng(2024)
    kinds = ['normal', 'planted', 'vqgan_init', 'int', 'unit']
    for trial in range(40):
        N = int(g.integers(1, 600)) if trial % 4 else int(g.integers(600, 2500))
        K = int(g.integers(1, 900)) if trial % 3 else int(g.integers(900, 3000))
        D = int(g.choice([8, 16, 24, 32, 40, 64, 96, 128, 200, 256, 320, 512, 520, 768, 1000]))
        metric = 'L2' if g.random() < 0.6 else 'Cosine'
        kind = kinds[int(g.integers(0, len(kinds)))]
        x, w = synth.make_inputs(kind, 500 + trial, N, K, D)
        scale = float(10.0 ** g.integers(-3, 4))
        x, w = (x * np.float32(scale)).astype(np.float32), (w * np.float32(scale)).astype(np.float32)
        bf16 = g.random() < 0.3
        if bf16:
            x = synth.bf16_round(x)
        if metric == 'Cosine':
            ref = co.cos_argmin(x, w)
            xq = ops.normalize_rows(dev(x, torch.bfloat16 if bf16 else None))
        else:
            ref = co.l2_argmin(x, w)
            xq = dev(x, torch.bfloat16 if bf16 else None)
        cb = ops.prepare_codebook(dev(w), metric)
        got = ops.argmin(xq, cb).cpu().numpy()
        assert np.array_equal(got, ref), f'trial {trial}: N={N} K={K} D={D} {metric} {kind} scale={scale} bf16={bf16}'


@pytest.mark.parametrize('N,K,D,metric', [(20000, 8192, 32, 'Cosine'), (16500, 1000, 32, 'Cosine'), (40000, 777, 24, 'L2'),
                                          (70001, 4099, 8, 'L2'), (33333, 300, 16, 'Cosine')])
def test_small_d_kernel_forms_agree(ops, N, K, D, metric):
    """D <= 32 at >= 16 384 tokens runs the group-record form (identification requests served by identify32_kernel, or the
    in-kernel replay of the 16x16x32 form), without aux reads for cosine
    (a codebook that does not fill its last stage reads them there) and with balanced workgroup sizes: every
    combination of the three knobs (8, 9, 10) returns the indices of the all-fp32 route, on the CPU oracle's definition."""
    from vector_quantization_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(N + K + D)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = torch.randn(N, D, device='cuda', generator=g)
    x[: N // 3] = w[torch.randint(0, K, (N // 3,), device='cuda', generator=g)] + 0.01 * x[: N // 3]   # near codes
    w[K // 2] = w[0]                                                                                       # an exact duplicate
    if metric == 'Cosine':
        xq, wq = ops.normalize_rows(x), ops.normalize_rows(w)
    else:
        xq, wq = x, w
    ref = ops.argmin_exact(xq, wq, metric)
    sample = slice(0, 600)
    oracle = (co.cos_argmin if metric == 'Cosine' else co.l2_argmin)(x[sample].cpu().numpy(), w.cpu().numpy())
    np.testing.assert_array_equal(ref[sample].cpu().numpy(), oracle)
    try:
        for noaux in (0, 1):
            for groups in (0, 1):
                for balance in (0, 1):
                    L.vqhip_set_tuning(8, noaux); L.vqhip_set_tuning(9, groups); L.vqhip_set_tuning(10, balance)
                    got = ops.argmin(xq, ops.prepare_codebook(w, metric))
                    assert torch.equal(got, ref), (noaux, groups, balance)
    finally:
        for key in (8, 9, 10):
            L.vqhip_set_tuning(key, 1)


@pytest.mark.parametrize('N,K,D,metric,hot', [(40000, 8192, 32, 'Cosine', 1), (50000, 16384, 8, 'L2', 1), (60000, 8192, 16, 'Cosine', 40)])
def test_group_request_lists_overflow_to_the_second_pass(ops, N, K, D, metric, hot):
    """The D <= 32 group path files one identification request per (token, lane half) under the group of the lane's best
    code tile; a bucket holds a fixed share of the request pool.  Latents crowded onto `hot` codes overflow those lists by an
    order of magnitude: the overflowing groups stay bounds, the rows take the second proposal pass, and the indices are
    still those of the all-fp32 route (the CPU oracle's definition on a row sample)."""
    g = torch.Generator(device='cuda').manual_seed(N + K + D + hot)
    w = torch.randn(K, D, device='cuda', generator=g)
    codes = torch.randint(0, K, (hot,), device='cuda', generator=g)
    x = w[codes[torch.randint(0, hot, (N,), device='cuda', generator=g)]] + 0.05 * torch.randn(N, D, device='cuda', generator=g)
    if metric == 'Cosine':
        xq, wq = ops.normalize_rows(x), ops.normalize_rows(w)
    else:
        x, w = ops.normalize_rows(x), ops.normalize_rows(w)          # the LlamaGen form: constant-norm codebook
        xq, wq = x, w
    ref = ops.argmin_exact(xq, wq, metric)
    got, st = ops.argmin(xq, ops.prepare_codebook(w, metric), return_stats=True)
    assert torch.equal(got, ref)
    sample = slice(0, 400)
    oracle = (co.cos_argmin if metric == 'Cosine' else co.l2_argmin)(x[sample].cpu().numpy(), w.cpu().numpy())
    np.testing.assert_array_equal(got[sample].cpu().numpy(), oracle)
    if hot == 1:      # most rows could not get a request in: second pass (and from there, lists overflowing, the fp32 pass)
        assert int(st[0]) + int(st[2]) > N // 2, st


@pytest.mark.parametrize('N,K,D,metric,dtype', [
    (3072, 16384, 256, 'Cosine', torch.float32),      # the CVQ-VAE step's shape
    (3072, 16384, 256, 'L2', torch.float32),
    (500, 1000, 24, 'L2', torch.float32),             # D % 32 != 0: a tail block; K % 64 != 0: a ragged code tile
    (700, 5000, 200, 'Cosine', torch.float32),
    (900, 8192, 768, 'L2', torch.float32),            # D > 256: the e-tile ring is refilled as it is consumed
    (640, 4099, 1024, 'L2', torch.bfloat16),          # bf16 rows, the largest D of the few-rows form
    (17, 777, 8, 'L2', torch.float32),
])
def test_last_resort_pass_both_forms(ops, N, K, D, metric, dtype):
    """The whole-codebook fp32 pass over listed rows has two forms chosen on the device by the length of the list — up to 16
    rows the VALU form (a lane per code, v_pk_fma_f32 chains), longer lists the MFMA form.  vqhip_set_tuning key 12 sends
    the first V rows of the batch through it whatever the earlier stages decided: the indices must stay those of the
    all-fp32 route, for list lengths on both sides of the switch, odd lengths (a half-empty row pair) and a single row."""
    from vector_quantization_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(N + K + D)
    w = torch.randn(K, D, device='cuda', generator=g)
    x = w[torch.randint(0, K, (N,), device='cuda', generator=g)] + 0.3 * torch.randn(N, D, device='cuda', generator=g)
    x[1] = w[7]                                        # an exact hit and a duplicated code: ties go to the lowest index
    w[K - 1] = w[7]
    if metric == 'Cosine':
        xq, wq = ops.normalize_rows(x), ops.normalize_rows(w)
    else:
        xq, wq = x, w
    xq = xq.to(dtype)
    ref = ops.argmin_exact(xq, wq, metric)
    cb = ops.prepare_codebook(w, metric)
    try:
        for V in (1, 2, 5, 12, 15, 16, 17, 40):
            L.vqhip_set_tuning(12, V)
            got, st = ops.argmin(xq, cb, return_stats=True)
            assert int(st[2]) >= min(V, N), (V, st)
            assert torch.equal(got, ref), (V, int((got != ref).sum()))
    finally:
        L.vqhip_set_tuning(12, 0)


@pytest.mark.parametrize('kind,metric,scale', [('normal', 'L2', 1.0), ('normal', 'Cosine', 1.0), ('vqgan_init', 'L2', 1.0),
                                                ('normal', 'L2', 1e-3), ('normal', 'L2', 300.0), ('planted', 'L2', 1.0)])
def test_margin_holds(ops, kind, metric, scale, D=256):
    """The rigorous error bound behind the exactness claim, checked directly: for EVERY (row, code) pair the fp16-MFMA
    proposal score is within margin/2 of the exact (float64) score, with slack to spare."""
    N, K = 256, 2048
    x, w = synth.make_inputs(kind, 31, N, K, D)
    x, w = (x * np.float32(scale)).astype(np.float32), (w * np.float32(scale)).astype(np.float32)
    if metric == 'Cosine':
        xe, we = co.normalize_rows(x), co.normalize_rows(w)
        en = np.zeros(K, np.float64)
        xq = dev(xe)
    else:
        xe, we = x, w
        en = co.row_sqnorm(w).astype(np.float64)
        xq = dev(x)
    cb = ops.prepare_codebook(dev(w), metric)
    scores, margin, se = ops.debug_proposal_scores(xq, cb)
    scores, margin, se = scores.cpu().numpy().astype(np.float64), margin.cpu().numpy().astype(np.float64), float(se.item())
    exact = se * (xe.astype(np.float64) @ we.astype(np.float64).T - 0.5 * en[None, :])
    err = np.abs(scores - exact).max(1)
    assert (margin > 0).all()
    ratio = err / (0.5 * margin)
    assert ratio.max() <= 1.0, f'error exceeds the bound: max ratio {ratio.max():.3f}'
    print(f'{kind}/{metric}/x{scale}: max |score error| / (margin/2) = {ratio.max():.4f} (median {np.median(ratio):.4f})')


@pytest.mark.parametrize('D', [32, 768, 1024])
def test_margin_holds_other_dims(ops, D):
    test_margin_holds(ops, 'normal', 'L2', 1.0, D=D)
    test_margin_holds(ops, 'normal', 'Cosine', 1.0, D=D)


def test_cluster_shape_d768(ops):
    """configs/cluster (CLIP/DINO/MAE/ViT features, D=768, K=8192, cosine, NearestAnchor): proposal path at 48 k-steps,
    row and column argmin against the oracle."""
    N, K, D = 1024, 8192, 768
    x, w = synth.make_inputs('normal', 768, N, K, D)
    xo, wo = co.normalize_rows(x), co.normalize_rows(w)
    cb = ops.prepare_codebook(dev(w), 'Cosine')
    idx = ops.argmin(dev(xo), cb)
    np.testing.assert_array_equal(idx.cpu().numpy(), co.cos_argmin(x, w))
    col = ops.col_argmin(dev(xo), dev(wo), 'Cosine')
    np.testing.assert_array_equal(col.cpu().numpy(), co.col_argmin(co.cos_dist(x, w)))


@pytest.mark.parametrize('D,K', [(64, 4096), (256, 2048), (768, 1024)])
def test_near_duplicate_codebook_floods_second_pass(ops, D, K):
    """Every code has three near-copies (relative perturbation 1e-3): most rows have >= 4 near-tied candidates, more
    than the proposal pass can identify, so thousands of rows take the packed second proposal pass (several row
    blocks and several items per workgroup) and the list re-rank.  Indices stay exact."""
    N = 20000
    rng = np.random.default_rng(4242 + D)
    base = rng.standard_normal((K // 4, D)).astype(np.float32)
    w = np.concatenate([base * np.float32(1.0 + 1e-3 * i) for i in range(4)], 0)
    w = w[rng.permutation(K)].copy()
    x = (base[rng.integers(0, K // 4, N)] + 0.05 * rng.standard_normal((N, D))).astype(np.float32)
    cb = ops.prepare_codebook(dev(w), 'L2')
    idx, st = ops.argmin(dev(x), cb, return_stats=True)
    assert int(st[0]) > 2000, f'expected a flooded second pass, got {int(st[0])} rows'
    np.testing.assert_array_equal(idx.cpu().numpy(), co.l2_argmin(x, w))


def _ordered_sum_reference(rows_of, idx, K, D):
    """The ordered route's association restated sequentially: the stable code-sorted token order is cut into ranges of
    64 positions; inside a range rows are added in order; a code spanning ranges is its range pieces added in range
    order (include/vqhip.h).  rows_of(n, k) -> fp32 row contributed by token n to code k."""
    order = np.argsort(idx, kind='stable')
    out = np.zeros((K, D), np.float32)
    pieces = {}
    for r0 in range(0, len(order), 64):
        cur, acc = None, None
        for n in order[r0:r0 + 64]:
            k = int(idx[n])
            if k != cur:
                if cur is not None:
                    pieces.setdefault(cur, []).append(acc)
                cur, acc = k, np.zeros(D, np.float32)
            acc = acc + rows_of(n, k)
        if cur is not None:
            pieces.setdefault(cur, []).append(acc)
    for k, ps in pieces.items():
        acc = ps[0]
        for q in ps[1:]:
            acc = acc + q
        out[k] = acc
    return order, out


@pytest.mark.parametrize('N,K,D', [(5000, 300, 64), (1, 1, 8), (1024, 16384, 256), (3001, 7, 32), (70000, 4096, 16)])
def test_ordered_codebook_sums(ops, N, K, D):
    """SURVEY.md §7 hard part 9: the ordered route — stable counting sort of the tokens by code, then sums in a fixed
    order — is integer-exact for the order and bit-exact for the sums against a sequential numpy restatement, and
    agrees with the atomic route within rounding."""
    g = synth.rng(77 + N)
    idx = g.integers(0, K, N)
    if K > 4:
        idx[idx == 3] = 2                                   # an unused code among used ones
    src = g.standard_normal((N, D), dtype=np.float32)
    counts, offsets, order = ops.token_order(dev(idx.astype(np.int64)), K)
    ref_order, ref = _ordered_sum_reference(lambda n, k: src[n], idx, K, D)
    np.testing.assert_array_equal(order.cpu().numpy(), ref_order)
    np.testing.assert_array_equal(counts.cpu().numpy(), np.bincount(idx, minlength=K))
    np.testing.assert_array_equal(offsets.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=K))]))
    got = ops.scatter_add_rows(dev(src), dev(idx.astype(np.int64)), K, ordered=True).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    atom = ops.scatter_add_rows(dev(src), dev(idx.astype(np.int64)), K, ordered=False).cpu().numpy()
    np.testing.assert_allclose(atom, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', [None, torch.bfloat16])
def test_ordered_codebook_gradient(ops, dtype):
    """grad_W of the fused backward on the ordered route: bit-equal to the sequential restatement of
    sum_n kw*(e_k - x_n), bit-reproducible run to run, and within rounding of the atomic route.  The planted inputs
    give codes with a handful of tokens and, through the repeated block, one code with hundreds (several ranges)."""
    N, K, D = 6000, 512, 64
    x, w = synth.make_inputs('planted', 9, N, K, D)
    x[1000:1700] = x[1000] + np.float32(1e-3) * synth.normal(3, 700, D)      # 700 tokens on one code
    if dtype is not None:
        x = synth.bf16_round(x)
    idx = co.l2_argmin(x, w)
    assert np.bincount(idx).max() >= 700
    g_cb = np.float32(0.7)
    xd, wd, idd = dev(x, dtype), dev(w), dev(idx.astype(np.int64))
    gcb = torch.tensor(float(g_cb), device='cuda')
    gz = dev(synth.normal(5, N, D))
    gx1, gw1 = ops.vq_backward(xd, wd, idd, gz, gcb, gcb, True, True, ordered=True)
    gx2, gw2 = ops.vq_backward(xd, wd, idd, gz, gcb, gcb, True, True, ordered=True)
    assert torch.equal(gw1, gw2) and torch.equal(gx1, gx2)
    gx3, gw3 = ops.vq_backward(xd, wd, idd, gz, gcb, gcb, True, True, ordered=False)
    assert torch.equal(gx1, gx3)
    np.testing.assert_allclose(gw1.cpu().numpy(), gw3.cpu().numpy(), rtol=2e-4, atol=1e-7)
    kw = np.float32(g_cb * np.float32(np.float32(2.0) / (np.float32(N) * np.float32(D))))
    _, ref = _ordered_sum_reference(lambda n, k: kw * (w[k] - x[n]), idx, K, D)
    np.testing.assert_array_equal(gw1.cpu().numpy(), ref)


def test_c_abi_standalone_program():
    """examples/abi_argmin.cpp: a C++ program that links libvqhip.so directly (HIP runtime only: no Python, no torch in
    the process) — compiled with hipcc on the GPU box, it quantizes, cross-checks against the fp32 entry point and a host
    nearest-neighbour, and reports the path counters."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    libdir = os.path.join(root, 'vector_quantization_amd')
    exe = os.path.join('/tmp', f'abi_argmin_{os.getpid()}')
    build = subprocess.run([hipcc, '--offload-arch=gfx950', '-I' + os.path.join(root, 'include'),
                            os.path.join(root, 'examples', 'abi_argmin.cpp'), '-L' + libdir, '-lvqhip',
                            '-Wl,-rpath,' + libdir, '-o', exe], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(run.stdout)
    assert run.returncode == 0 and 'ABI OK' in run.stdout, run.stdout + run.stderr
