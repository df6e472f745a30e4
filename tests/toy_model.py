"""A stand-in for the reference's VQ models around the drop-in quantizer (vq/tasks/image_tokenization/models/base.py:116-128:
encoder -> post_encode -> 'b c h w -> (b h w) c' -> quantizer -> '(b h w) c -> b c h w' -> pre_decode -> decoder): a 1x1 conv
either side.  The module name `_quantizer` is the reference's (configs/vqgan/runner.py:47 selects it by that name).  Used by
the DDP / FSDP tests — what every shipped training config wraps its model in (configs/strategies/ddp.py:5-6, fsdp.py:5-8)."""
import torch
from torch import nn

EMB = 'torch_nn_modules_sparse_Embedding'


def quantizer_cfg(kind, K, D):
    emb = dict(type=EMB, num_embeddings=K, embedding_dim=D)
    if kind == 'vqkd':       # configs/vqkd/model.py:20-26
        return dict(type='VQKDQuantizer', embedding=emb, distance=dict(type='CosineDistance'),
                    callbacks=[dict(type='VQKDCallback', ema=dict())],
                    losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))
    cfg = dict(type='VQGANQuantizer', embedding=emb, distance=dict(type='L2Distance' if kind == 'vqgan' else 'CosineDistance'),
               losses=dict(vqgan_loss=dict(type='VQGANLoss')))           # configs/vqgan/model.py:19-23
    if kind == 'cvq':        # configs/cvqvae/quantizer.py:1-6
        cfg['callbacks'] = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]
    return cfg


class ToyVQModel(nn.Module):

    def __init__(self, quantizer: nn.Module, channels: int = 8) -> None:
        super().__init__()
        d = quantizer.embedding_dim
        self._post_encode = nn.Conv2d(channels, d, 1)
        self._quantizer = quantizer
        self._pre_decode = nn.Conv2d(d, channels, 1)
        self.quant_call = None            # optional replacement of the quantizer call (graphs.GraphedQuantizer)

    def forward(self, image: torch.Tensor):
        h = self._post_encode(image)
        b, d, hh, ww = h.shape
        x = h.permute(0, 2, 3, 1).reshape(-1, d)
        if self.quant_call is not None:
            z, q_loss, quant = self.quant_call(x)
        else:
            z, q_loss, memo = self._quantizer(x, {})
            quant = memo['quant']
        z_map = z.reshape(b, hh, ww, d).permute(0, 3, 1, 2)
        out = self._pre_decode(z_map.to(h.dtype))
        return out, q_loss, quant


def build_toy(kind: str, K: int, D: int, weight: torch.Tensor, device, channels: int = 8, seed: int = 0) -> ToyVQModel:
    from vector_quantization_amd import Config, build_quantizer
    torch.manual_seed(seed)                # identical initialisation on every rank (what DDP's first broadcast establishes)
    q = build_quantizer(quantizer_cfg(kind, K, D))
    q.train()
    q.init_weights(Config(dict(type='vqgan') if kind != 'vqkd' else {}))
    q._forward_pre_hooks.clear()
    model = ToyVQModel(q, channels).to(device)
    with torch.no_grad():
        model._quantizer.embedding.weight.copy_(weight)
    if kind == 'vqkd':                     # configs/vqkd/model.py:76-82: no_grad on the quantizer's parameters
        for p in model._quantizer.parameters():
            p.requires_grad_(False)
    return model


def train_steps(model: nn.Module, images, lr: float = 0.05, autocast: bool = False, grads_out=None):
    """A few optimizer steps; returns the per-step (loss, tokens).  `model` may be the bare module or a DDP / FSDP wrapper."""
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=lr)
    rec = []
    for image in images:
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            out, q_loss, quant = model(image)
            loss = out.float().pow(2).mean() + q_loss
        loss.backward()
        if grads_out is not None:
            grads_out.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        opt.step()
        rec.append((loss.detach().clone(), quant.detach().clone()))
    return rec
