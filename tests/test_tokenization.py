"""rearrangement around the quantizer, on-disk token formats, codebook metrics."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co, synth


def test_token_file_formats_roundtrip(tmp_path):
    """File layouts of the reference's tokenize runners (CPU: pure I/O, no compute)."""
    from vector_quantization_amd import tokenization as T
    quant = torch.arange(2 * 4 * 4).reshape(-1)
    p = T.save_tokens(tmp_path, 7, ['a', 'b'], torch.tensor([3, 5]), quant, (2, 8, 4, 4), rank=1)
    assert p.name == '7_1.pth' and p.parent.name == 'tokens'
    tok = T.load_tokens(p)
    assert tok['id_'] == ['a', 'b'] and tok['tokens'].shape == (2, 4, 4) and tok['tokens'].dtype == torch.int64
    assert torch.equal(tok['tokens'].reshape(-1), quant) and torch.equal(tok['category'], torch.tensor([3, 5]))
    # LlamaGen: ten crops of one image per rank and iteration, i = (iter-1)*world+rank
    q10 = torch.arange(10 * 16)
    code, label = T.save_llamagen(tmp_path, 256, 3, q10, torch.tensor([417]), rank=2, world_size=8)
    assert code.name == '18.npy' and code.parent.name == 'imagenet256_codes' and label.parent.name == 'imagenet256_labels'
    c = np.load(code)
    assert c.shape == (1, 10, 16) and c.dtype == np.int64 and (c.reshape(-1) == q10.numpy()).all()
    assert np.load(label).tolist() == [417]


@pytest.mark.gpu
def test_rearrange_quantize_and_metrics(tmp_path):
    from vector_quantization_amd import build_quantizer, Config, ops
    from vector_quantization_amd import tokenization as T
    B, C, H, W, K = 3, 64, 16, 16, 512
    g = synth.rng(5)
    x = g.standard_normal((B, C, H, W), dtype=np.float32)
    w = g.standard_normal((K, C), dtype=np.float32)
    xd = torch.from_numpy(x).cuda()
    # transposes are pure data movement: bit-exact against numpy, both dtypes, ragged tile edges
    np.testing.assert_array_equal(T.to_tokens(xd).cpu().numpy(), x.transpose(0, 2, 3, 1).reshape(-1, C))
    xb = xd.bfloat16()
    assert torch.equal(T.to_tokens(xb), xb.permute(0, 2, 3, 1).reshape(-1, C))
    odd = torch.from_numpy(g.standard_normal((2, 37, 5, 13), dtype=np.float32)).cuda()
    assert torch.equal(T.to_map(T.to_tokens(odd), 2, 5, 13), odd)
    q = build_quantizer(dict(type='VQGANQuantizer',
                             embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=C),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
    q.init_weights(Config(type='vqgan'))
    q = q.cuda().eval()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(w))
    ref = co.l2_argmin(x.transpose(0, 2, 3, 1).reshape(-1, C), w)
    quant, memo = T.encode_to_quant(q, xd, {})
    assert quant.shape == (B, H, W) and memo['quantizer']['x_shape'] == xd.shape
    np.testing.assert_array_equal(quant.reshape(-1).cpu().numpy(), ref)
    z, _ = T.decode_from_quant(q, quant, {})
    np.testing.assert_array_equal(z.detach().cpu().numpy(), w[ref].reshape(B, H, W, C).transpose(0, 3, 1, 2))
    # model-level quantize with gradients flowing back to the BCHW feature map
    q.train()
    xg = xd.clone().requires_grad_(True)
    zq, loss, memo = T.quantize(q, xg, {})
    assert zq.shape == xd.shape
    (loss + zq.sum()).backward()
    assert xg.grad.shape == xd.shape and torch.isfinite(xg.grad).all()
    # saved file = what TokenizeCallback writes
    p = T.save_tokens(tmp_path, 1, ['i0', 'i1', 'i2'], torch.tensor([1, 2, 3]), memo['quantizer']['quant'],
                      memo['quantizer']['x_shape'])
    assert T.load_tokens(p)['tokens'].shape == (B, H, W)
    # metrics from accumulated counts vs the reference formulas
    cc = T.CodebookCounts(K)
    cc.update(quant)
    cc.update(quant)
    s = cc.summary()
    counts = torch.from_numpy(co.bincount(ref, K) * 2)
    assert abs(s['codebook_usage'] - counts.bool().sum().item() / K) < 1e-12
    ent = torch.distributions.Categorical(counts / counts.sum()).entropy().item()
    assert abs(s['codebook_ppl'] - ent) < 1e-5


def test_channels_last_map_is_the_token_matrix_zero_copy():
    """a channels-last latent map is rearranged to tokens and back without moving a byte (CPU
    tensors are enough: no kernel runs on this path)."""
    from vector_quantization_amd import tokenization as T
    B, C, H, W = 2, 12, 3, 5
    x = torch.randn(B, C, H, W).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert T.is_token_major(x) and not T.is_token_major(torch.randn(B, C, H, W))
    t = T.to_tokens(x)
    assert t.shape == (B * H * W, C) and t.data_ptr() == x.data_ptr() and t.is_contiguous()
    assert torch.equal(t, x.permute(0, 2, 3, 1).reshape(-1, C))
    m = T.to_map(t, B, H, W, token_major=True)
    assert m.shape == x.shape and m.data_ptr() == x.data_ptr() and torch.equal(m, x)
    assert m.is_contiguous(memory_format=torch.channels_last)
    m.square().sum().backward()                      # plain autograd through the views
    assert torch.allclose(x.grad, 2 * x.detach())


def test_conv_connector_mirror_builds_and_emits_token_major_maps():
    """connectors/conv.py: same config keys and state-dict; the 1x1 conv output is channels-last so the quantizer side
    gets its token matrix as a view.  Values equal the NCHW conv up to the GEMM summation order."""
    from vector_quantization_amd import Config
    from vector_quantization_amd import tokenization as T
    from vector_quantization_amd.connectors import BaseConnector, ConvConnector
    from vector_quantization_amd.registries import VQITConnectorRegistry
    torch.manual_seed(0)
    c = VQITConnectorRegistry.build(Config(type='ConvConnector'), in_channels=16, out_channels=8)
    assert isinstance(c, ConvConnector) and (c.in_channels, c.out_channels) == (16, 8)
    assert sorted(c.state_dict()) == ['_conv.bias', '_conv.weight'] and c.state_dict()['_conv.weight'].shape == (8, 16, 1, 1)
    x = torch.randn(2, 16, 4, 4)
    y, memo = c(x, {'k': 1})
    assert memo == {'k': 1} and y.shape == (2, 8, 4, 4) and T.is_token_major(y)
    ref = torch.nn.functional.conv2d(x, c._conv.weight, c._conv.bias)
    assert torch.allclose(y, ref, atol=1e-5)
    assert T.to_tokens(y).data_ptr() == y.data_ptr()
    ident = VQITConnectorRegistry.build(Config(type='BaseConnector'), in_channels=8, out_channels=8)
    assert isinstance(ident, BaseConnector) and ident(y, {})[0] is y
    plain = VQITConnectorRegistry.build(Config(type='ConvConnector', channels_last=False, conv=dict(kernel_size=3, padding=1)),
                                        in_channels=16, out_channels=8)
    assert plain(x, {})[0].is_contiguous() and plain._conv.kernel_size == (3, 3)


@pytest.mark.gpu
def test_quantize_token_major_equals_nchw():
    """The channels-last route (views only) and the NCHW route (HIP transposes) give identical z, loss, tokens and
    input gradients."""
    from vector_quantization_amd import build_quantizer, Config
    from vector_quantization_amd import tokenization as T
    B, C, H, W, K = 4, 64, 16, 16, 1024
    g = synth.rng(11)
    x = torch.from_numpy(g.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    q = build_quantizer(dict(type='VQGANQuantizer',
                             embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=C),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
    q.init_weights(Config(type='vqgan'))
    q = q.cuda().train()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g.standard_normal((K, C), dtype=np.float32)))
    outs = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        xi = x.clone().contiguous(memory_format=fmt).requires_grad_(True)
        q.zero_grad()
        z, loss, memo = T.quantize(q, xi, {})
        (loss + (z * z).sum()).backward()
        outs.append((z.detach().clone(), loss.detach().clone(), memo['quantizer']['quant'].clone(), xi.grad.clone(),
                     q.embedding.weight.grad.clone()))
        if fmt is torch.channels_last:
            assert T.is_token_major(z)
    for i, (a, b) in enumerate(list(zip(*outs))[:4]):
        if i == 1:        # the loss: the map kernel and the token kernel add the same squares in a different (double) order
            assert abs(a.item() - b.item()) <= 1e-6 * abs(b.item())
        else:
            assert torch.equal(a, b)
    assert torch.allclose(outs[0][4], outs[1][4], rtol=1e-5, atol=1e-6)     # codebook grad: atomic scatter-add order


@pytest.mark.gpu
def test_bulk_tokenization_loop_single_rank(tmp_path):
    """tools/tokenize_synthetic.py (BASELINE configs[4] in miniature) at world size 1: token files in the reference's
    layout, histogram-based metrics at the end."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('tokenize_synthetic', os.path.join(os.path.dirname(__file__), '..', 'tools', 'tokenize_synthetic.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    summary = mod.main(['--work-dir', str(tmp_path), '--iters', '2', '--images', '8', '--codes', '512', '--dim', '32'])
    files = sorted(p.name for p in (tmp_path / 'tokens').iterdir())
    assert files == ['1_0.pth', '2_0.pth']
    from vector_quantization_amd import tokenization as T
    tok = T.load_tokens(tmp_path / 'tokens' / '1_0.pth')
    assert tok['tokens'].shape == (8, 16, 16) and tok['tokens'].dtype == torch.int64 and len(tok['id_']) == 8
    assert 0.0 < summary['codebook_usage'] <= 1.0 and summary['codebook_ppl'] > 0.0


@pytest.mark.gpu
def test_connector_quantize_path_matches_reference():
    """f3 against the reference's own files (fixture: its ConvConnector, BaseModel.quantize / encode_to_quant and
    VQGANQuantizer): the product's connectors reproduce the 1x1 convs to GEMM rounding, and — fed the reference's latent
    map, NCHW or channels-last — `tokenization.quantize` returns the reference's tokens and straight-through map exactly
    and its loss to 1e-5."""
    import json
    from vector_quantization_amd import Config, build_quantizer, connectors as C, tokenization as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'connector_path.npz'))
    spec = json.loads(str(g['spec']))
    B, Cin, D, H, W, K = (spec[k] for k in ('B', 'Cin', 'D', 'H', 'W', 'K'))
    gen = synth.rng(spec['seed'])
    x_in = gen.standard_normal((B, Cin, H, W), dtype=np.float32)
    wcb = gen.standard_normal((K, D), dtype=np.float32)
    np.testing.assert_array_equal(x_in, g['x_in'])
    assert synth.sha(wcb) == str(g['w_sha'])
    post = C.VQITConnectorRegistry.build(dict(type='ConvConnector', in_channels=Cin, out_channels=D)).cuda()
    pre = C.VQITConnectorRegistry.build(dict(type='ConvConnector', in_channels=D, out_channels=Cin)).cuda()
    post.load_state_dict({'_conv.weight': torch.from_numpy(g['post_weight']), '_conv.bias': torch.from_numpy(g['post_bias'])})
    pre.load_state_dict({'_conv.weight': torch.from_numpy(g['pre_weight']), '_conv.bias': torch.from_numpy(g['pre_bias'])})
    q = build_quantizer(dict(type='VQGANQuantizer',
                             embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))
    q.init_weights(Config(type='vqgan'))
    q = q.cuda().eval()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(wcb))
        x_map, _ = post(torch.from_numpy(x_in).cuda(), {})
        assert T.is_token_major(x_map)                                             # the connector hands over the token matrix
        np.testing.assert_allclose(x_map.cpu().numpy(), g['x_map'], rtol=1e-5, atol=1e-5)
        ref_map = torch.from_numpy(g['x_map']).cuda()
        for fmt in (torch.contiguous_format, torch.channels_last):
            z_map, loss, memo = T.quantize(q, ref_map.contiguous(memory_format=fmt), {})
            np.testing.assert_array_equal(memo['quantizer']['quant'].reshape(B, H, W).cpu().numpy(), g['quant'].astype(np.int64))
            np.testing.assert_array_equal(z_map.cpu().numpy(), g['z_map'])          # x + (z - x): bit-exact
            assert abs(loss.item() - float(g['loss'])) <= 1e-5 * max(1.0, float(g['loss']))
            assert tuple(memo['quantizer']['x_shape']) == (B, D, H, W)
            quant, _ = T.encode_to_quant(q, ref_map.contiguous(memory_format=fmt), {})
            np.testing.assert_array_equal(quant.cpu().numpy(), g['quant'].astype(np.int64))
        out, _ = pre(torch.from_numpy(g['z_map']).cuda(), {})
        np.testing.assert_allclose(out.cpu().numpy(), g['out'], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('B,D,H,W,K,dtype', [(256, 256, 16, 16, 16384, torch.bfloat16),     # the bench's map: channels shared by 2 workgroups
                                             (3, 256, 16, 16, 4096, torch.float32),       # few tiles: channels shared by 8
                                             (2, 32, 32, 32, 1000, torch.float32),        # 4 tiles per image, one chunk
                                             (5, 96, 16, 16, 777, torch.bfloat16),        # 3 chunks: no sharing possible
                                             (1, 768, 32, 16, 512, torch.float32),
                                             (2, 64, 16, 16, 300, None)])                 # decode: no latents, no loss
def test_gather_ste_map_whole_image_tiles(B, D, H, W, K, dtype):
    """vqhip_gather_ste_map takes tiles of 256 consecutive positions of one image when the images hold a multiple of 256
    positions (gather_ste_map256_kernel: 1 KiB channel rows per wave-store); vqhip_set_tuning key 13 = 0 keeps the 64-token
    tiles.  Both must write the map of the token-major gather ('(b h w) c -> b c h w', models/base.py:126-127) bit for bit
    and the same loss to double-sum rounding."""
    from vector_quantization_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(B + D + H * W + K)
    N = B * H * W
    e = torch.randn(K, D, device='cuda', generator=g)
    idx = torch.randint(0, K, (N,), device='cuda', generator=g)
    x = None if dtype is None else torch.randn(N, D, device='cuda', generator=g).to(dtype)
    try:
        L.vqhip_set_tuning(13, 1)
        new_map, new_mse = ops.gather_ste_map(x, e, idx, B, H, W, beta=0.25)
        L.vqhip_set_tuning(13, 0)
        old_map, old_mse = ops.gather_ste_map(x, e, idx, B, H, W, beta=0.25)
    finally:
        L.vqhip_set_tuning(13, 1)
    assert torch.equal(new_map, old_map)
    if x is None:
        ref = e[idx]
    else:
        xf = x.float()
        ref = xf + (e[idx] - xf)
    ref_map = ref.reshape(B, H * W, D).permute(0, 2, 1).reshape(B, D, H, W)
    assert torch.equal(new_map, ref_map)
    if x is not None:
        want = ((e[idx].double() - x.double()) ** 2).mean().item()
        assert abs(new_mse[0].item() - want) <= 1e-6 * want and abs(old_mse[0].item() - want) <= 1e-6 * want
        assert abs(new_mse[2].item() - (want + 0.25 * want)) <= 2e-6 * want


@pytest.mark.gpu
@pytest.mark.parametrize('B,D,H,W,K,dist,dtype', [(4, 256, 16, 16, 4096, 'L2', torch.bfloat16), (3, 32, 14, 14, 1000, 'Cosine', torch.float32),
                                                  (3, 64, 32, 16, 1024, 'L2', torch.float32), (2, 256, 16, 16, 2048, 'Cosine', torch.bfloat16),
                                                  (2, 8, 16, 16, 2048, 'L2', torch.float32), (5, 64, 7, 9, 777, 'L2', torch.float32),
                                                  (2, 768, 14, 14, 512, 'Cosine', torch.bfloat16)])
def test_nchw_map_route_runs_no_transpose_and_matches_token_route(B, D, H, W, K, dist, dtype, monkeypatch):
    """f3 as SURVEY.md §8f row 3 states it: handed an NCHW-contiguous map, `quantize` / `encode_to_quant` /
    `decode_from_quant` read and write the map directly — 'b c h w -> (b h w) c' inside the encode's first kernel
    (vqhip_encode_map), '(b h w) c -> b c h w' inside the gather (vqhip_gather_ste_map) — and return what the token route
    (explicit rearrangement, models/base.py:124-127) returns: identical tokens, identical straight-through map, loss to 1e-6,
    in eval and in train mode (CVQ-VAE callback included), with gradients reaching the map."""
    from vector_quantization_amd import Config, build_quantizer, ops
    from vector_quantization_amd import tokenization as T
    g = torch.Generator(device='cuda').manual_seed(B * D + H)
    x = torch.randn(B, D, H, W, device='cuda', generator=g).to(dtype)
    wcb = torch.randn(K, D, device='cuda', generator=g)
    cbs = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))] if dist == 'Cosine' else []

    def make(train):
        q = build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                                 distance=dict(type=f'{dist}Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')), callbacks=cbs))
        q.train(train)
        q.init_weights(Config(type='vqgan'))
        q = q.cuda()
        with torch.no_grad():
            q.embedding.weight.copy_(wcb)
        return q

    x_tok = x.permute(0, 2, 3, 1).reshape(-1, D).contiguous()          # the reference's rearrangement, done by torch here
    real_transpose = ops.transpose_last2

    def no_transpose(*a, **k):
        raise AssertionError('a transpose kernel ran on the NCHW map route')

    # ---- eval: tokens, map, loss; decode of image-shaped tokens straight into the map -------------------------------------
    q = make(False)
    with torch.no_grad():
        z_ref, loss_ref, memo_ref = q(x_tok, {})
        monkeypatch.setattr(ops, 'transpose_last2', no_transpose)
        z_map, loss, memo = T.quantize(q, x, {})
        quant, _ = T.encode_to_quant(q, x, {})
        z_dec, _ = T.decode_from_quant(q, quant, {})
        monkeypatch.setattr(ops, 'transpose_last2', real_transpose)
    assert z_map.shape == (B, D, H, W) and z_map.is_contiguous() and quant.shape == (B, H, W)
    assert torch.equal(memo['quantizer']['quant'], memo_ref['quant']) and torch.equal(quant.reshape(-1), memo_ref['quant'])
    assert torch.equal(z_map, z_ref.reshape(B, H, W, D).permute(0, 3, 1, 2))
    assert abs(loss.item() - loss_ref.item()) <= 1e-6 * max(1e-6, abs(loss_ref.item()))
    assert torch.equal(memo['quantizer']['x'], x_tok) and tuple(memo['quantizer']['x_shape']) == (B, D, H, W)
    assert torch.equal(z_dec, wcb[memo_ref['quant']].reshape(B, H, W, D).permute(0, 3, 1, 2))
    # ---- train: codebook update by the callback, gradients to the map and the codebook ----------------------------------------
    up = torch.randn(B, D, H, W, device='cuda', generator=g)
    q1, q2 = make(True), make(True)
    xt = x_tok.clone().requires_grad_(True)
    z1, l1, m1 = q1(xt, {})
    (l1 + (z1 * up.permute(0, 2, 3, 1).reshape(-1, D)).sum()).backward()
    xm = x.clone().requires_grad_(True)
    if ops.backward_map_supported(D, H * W):           # the backward reads and writes the map as well (vqhip_vq_backward_map)
        monkeypatch.setattr(ops, 'transpose_last2', no_transpose)
    z2, l2, m2 = T.quantize(q2, xm, {})
    (l2 + (z2 * up).sum()).backward()
    monkeypatch.setattr(ops, 'transpose_last2', real_transpose)
    assert torch.equal(m2['quantizer']['quant'], m1['quant']) and torch.equal(q1.embedding.weight.detach(), q2.embedding.weight.detach())
    assert torch.equal(z2, z1.detach().reshape(B, H, W, D).permute(0, 3, 1, 2))
    assert abs(l1.item() - l2.item()) <= 1e-6 * max(1e-6, abs(l1.item()))
    assert xm.grad.dtype == dtype and torch.equal(xm.grad, xt.grad.reshape(B, H, W, D).permute(0, 3, 1, 2))
    if q1.embedding.weight.grad is not None:
        assert torch.allclose(q1.embedding.weight.grad, q2.embedding.weight.grad, rtol=1e-5, atol=1e-7)
