"""rearrangement around the quantizer, on-disk token formats, codebook metrics."""
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
