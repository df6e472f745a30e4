"""VectorQuantizer / VQGANQuantizer / VQKDQuantizer — mirror of vq/algorithms/vq/quantizers.py:19-117,
vq/algorithms/vqgan/quantizer.py:11-21 and vq/algorithms/vqkd/quantizers/base.py:11-15."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from .. import functional as VF
from .. import ops
from ..config import Config, Item, RegistryMeta
from ..registries import InitRegistry, ModelRegistry, VQITQuantizerDistanceRegistry, VQITQuantizerRegistry
from .memo import Memo, get_memo
from .quantizer_api import BaseQuantizer
from .distances import BaseDistance, LazyDistance
from .losses import CodebookLoss, CommitmentLoss, VQGANLoss


@VQITQuantizerRegistry.register_()
class VectorQuantizer(BaseQuantizer):
    # False (or VQHIP_ONE_CALL=0 in the environment): the callback-driven training forwards run hook by hook, one library call
    # per piece, instead of as ONE call (train_step.py) — the same values either way (tests/test_gpu_one_call.py)
    one_call_steps = os.environ.get('VQHIP_ONE_CALL', '1') != '0'

    def __init__(self, *args, embedding: nn.Embedding, distance: BaseDistance, fused: bool = True,
                 cache_codebook: bool = False, **kwargs) -> None:
        """``fused`` / ``cache_codebook`` are extensions (defaults keep the reference semantics):
        fused          — decode + STE + plain MSE losses in one kernel when no callback customises decode/loss;
        cache_codebook — reuse the prepared codebook image while ``weight`` is bit-for-bit unchanged
                         (opt-in for frozen-codebook tokenisation; callers that mutate the weight must call
                         ``invalidate_codebook()``)."""
        super().__init__(*args, **kwargs)
        self._embedding = embedding
        self._distance = distance
        self._fused = fused
        self._cache_codebook = cache_codebook
        self._prepared: Optional[ops.PreparedCodebook] = None
        self._prepared_key = None

    @classmethod
    def embedding_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config.embedding = ModelRegistry.build_or_return(config.embedding)
        return config

    @classmethod
    def distance_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config.distance = VQITQuantizerDistanceRegistry.build_or_return(config.distance)
        return config

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        config = cls.embedding_build_pre_hook(config, registry, item)
        config = cls.distance_build_pre_hook(config, registry, item)
        return config

    @property
    def embedding(self) -> nn.Embedding:
        return self._embedding

    @property
    def distance(self) -> BaseDistance:
        return self._distance

    @property
    def embedding_dim(self) -> int:
        return self._embedding.embedding_dim

    @property
    def codebook_size(self) -> int:
        return self._embedding.num_embeddings

    @property
    def embeddings(self) -> torch.Tensor:
        return self._embedding.weight.detach().clone()

    def _init_weights(self, config: Config) -> bool:
        config = Config(config)
        if 'type' not in config:            # nothing configured: keep nn.Embedding's own initialisation
            return False
        func = InitRegistry.resolve(config.pop('type'))(**config)
        with torch.no_grad():
            func(self._embedding.weight)
        self.invalidate_codebook()
        return False

    # ---- encode: fused distance + argmin -------------------------------------------------------------------------
    def invalidate_codebook(self) -> None:
        self._prepared = None
        self._prepared_key = None

    def _prepare(self, w: torch.Tensor) -> ops.PreparedCodebook:
        if not self._cache_codebook:
            return self._distance.prepare(w)
        key = (w.data_ptr(), w._version, tuple(w.shape), w.device, self._distance.metric)
        if self._prepared is None or self._prepared_key != key:
            self._prepared = self._distance.prepare(w)
            self._prepared_key = key
        return self._prepared

    def _encode(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        """quantizers.py:92-100.  The reference clones the weight, materialises d[N, K] and takes argmin; here the
        weight is read in place, memo['distance'] is lazy, and the code-hit histogram falls out of the epilogue."""
        w = self._embedding.weight.detach()
        shape = x.shape[:-1]
        x2 = x.detach().reshape(-1, x.shape[-1])
        # the code-hit histogram is a by-product of the re-rank epilogue; only the training callbacks consume it
        hist = None
        want_hist = self.training and len(self._callbacks.callbacks) > 0
        stash = {}
        if self._cache_codebook:
            if want_hist:
                hist = torch.zeros(self.codebook_size, dtype=torch.int32, device=x.device)
            quant = self._distance.argmin(x2, w, hist=hist, prepared=self._prepare(w), stash=stash)
        else:                                   # the codebook may have changed since the last call: image made in the same call
            if want_hist:                       # zeroed by the call's first launch: no separate fill kernel
                hist = torch.empty(self.codebook_size, dtype=torch.int32, device=x.device)
            quant = self._distance.encode(x2, w, hist=hist, stash=stash, zero_hist=True)
        # memo['distance'] stays symbolic.  With autograd on, its operands keep their graph (EntropyLoss differentiates
        # through the matrix: losses.py:130-153); the codebook operand is an alias of the CURRENT weight storage, so the
        # values are those of encode time even after a callback rebinds weight.data (the reference clones them: :97).
        if torch.is_grad_enabled() and (x.requires_grad or self._embedding.weight.requires_grad):
            weight = self._embedding.weight
            memo['distance'] = LazyDistance(self._distance, x.reshape(-1, x.shape[-1]), weight.view_as(weight),
                                            xq=stash.get('xq'), eq=stash.get('eq'), metric=stash.get('metric'))
        else:
            memo['distance'] = LazyDistance(self._distance, x2, w, xq=stash.get('xq'), eq=stash.get('eq'),
                                            metric=stash.get('metric'))
        if hist is not None:
            memo['hist'] = hist
        return quant.reshape(shape), memo

    def _decode(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        z = VF.embedding(self._embedding.weight, quant)
        return z, memo

    # ---- forward ---------------------------------------------------------------------------------------------------
    def _fusable(self) -> bool:
        if not self._fused or self._callbacks.overrides_decode_or_loss():
            return False
        if type(self)._decode is not VectorQuantizer._decode or type(self)._loss is not BaseQuantizer._loss:
            return False
        for loss in self._losses.values():
            if not (isinstance(loss, (VQGANLoss, CodebookLoss, CommitmentLoss)) and loss.plain):
                return False
        return True

    def _loss_values(self, memo: Memo, m_cb, m_cm, m_vqgan, betas, like: torch.Tensor) -> torch.Tensor:
        """memo['loss'][name] of every configured (plain MSE) loss from the three values the fused kernel finished, and
        their fp32 sum (base.py:151-171)."""
        losses = {}
        for name, loss in self._losses.items():
            if isinstance(loss, VQGANLoss):
                # codebook + beta * commitment: finished inside the gather kernel for the (one) VQGANLoss of the shipped
                # configs; a second VQGANLoss with another beta takes the two-term form
                losses[name] = m_vqgan if loss.beta == betas[0] else torch.add(m_cb, m_cm, alpha=loss.beta)
            elif isinstance(loss, CodebookLoss):
                losses[name] = m_cb
            else:
                losses[name] = m_cm
        loss_memo = get_memo(memo, 'loss')
        loss_memo.update(losses)
        memo['loss'] = loss_memo
        values = list(losses.values())
        if len(values) == 1:                  # 0 + v == v: skip the zero-fill and the add of the general form below
            return values[0]
        return sum(values, like.new_zeros([], dtype=torch.float32))

    def _one_call_step(self, x: torch.Tensor):
        """The training forwards that ONE library call enqueues (train_step.py): a VQGAN-style quantizer whose only callback
        is a CVQVAECallback in its sparse-anchor flow, or a VQ-KD quantizer (VQKDCallback + CommitmentLoss with norm=True:
        configs/vqkd/model.py:20-26).  None: the step runs hook by hook as below."""
        if not (self.one_call_steps and self._fused and x.dim() == 2):
            return None
        cbs = self._callbacks.callbacks
        if len(cbs) > 1 or type(self)._encode is not VectorQuantizer._encode or type(self)._decode is not VectorQuantizer._decode \
                or type(self)._loss is not BaseQuantizer._loss or type(self).encode is not BaseQuantizer.encode:
            return None
        from .callbacks import CVQVAECallback, NormalizeCallback, VQKDCallback
        from .distances import CosineDistance, L2Distance
        if len(cbs) == 0 or type(cbs[0]) is NormalizeCallback:
            # no update callback (VQGAN: configs/vqgan/model.py:19-23), or NormalizeCallback alone (LlamaGen: configs/llamagen/
            # vqgan.py:18-20) — train and eval alike: vqhip_vq_forward
            w = self._embedding.weight
            ok = (self._fusable() and not self._cache_codebook and type(self._distance) in (L2Distance, CosineDistance)
                  and x.is_cuda and 0 < x.shape[0] < (1 << 31) and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous())
            return self._forward_plain if ok else None
        if not self.training:
            return None
        cb = cbs[0]
        if type(cb) is CVQVAECallback:
            return self._forward_cvq if (self._fusable() and cb.fused_forward_ok(x)) else None
        if type(cb) is VQKDCallback:
            losses = list(self._losses.values())
            ok = (len(losses) == 1 and type(losses[0]) is CommitmentLoss and losses[0]._mse.norm
                  and losses[0]._mse._weight._value == 1.0 and losses[0]._weight._value == 1.0)
            return self._forward_vqkd if (ok and cb.fused_forward_ok(x)) else None
        return None

    def _forward_cvq(self, x: torch.Tensor, memo: Memo):
        cb = self._callbacks.callbacks[0]
        betas = [loss.beta for loss in self._losses.values() if isinstance(loss, VQGANLoss)]
        quant, z_ste, m_cb, m_cm, m_vqgan = cb.fused_forward(x, memo, betas[0] if betas else 0.0)
        memo.update(x=x, quant=quant)
        memo['decode'] = get_memo(memo, 'decode')
        return z_ste, self._loss_values(memo, m_cb, m_cm, m_vqgan, betas, x), memo

    def _forward_plain(self, x: torch.Tensor, memo: Memo):
        """encode (+ NormalizeCallback.before_encode) + decode + MSE losses + STE from one library call (train_step.vq_forward)."""
        from .. import train_step
        cbs = self._callbacks.callbacks
        weight = self._embedding.weight
        D = weight.shape[1]
        normalize = len(cbs) == 1
        w_in = weight.detach()
        w_out = None
        if normalize:
            w_out = w_in if cbs[0]._writes_in_place(weight) else torch.empty_like(w_in)
        betas = [loss.beta for loss in self._losses.values() if isinstance(loss, VQGANLoss)]
        beta = betas[0] if betas else 0.0
        want_hist = self.training and len(cbs) > 0
        out = train_step.vq_forward(x.detach(), w_in, w_out, self._distance.metric_for(D), beta, normalize=normalize, want_hist=want_hist)
        if normalize:
            from ..utils import Store, is_sync
            if Store.DRY_RUN:
                assert is_sync(w_out)
            if w_out.data_ptr() != weight.data_ptr():
                weight.data = w_out                              # callbacks/update.py:56
            self.invalidate_codebook()
        done = VF._Computed(xn=out['xn'], z_ste=out['z_ste'], mse=out['mse'], idx=out['idx'])
        xn, z_ste, m_cb, m_cm, m_vqgan = VF.vq_step(x, weight, done, beta)
        rows = xn if normalize else x
        enc = get_memo(memo, 'encode')
        prepared = out['prepared']
        cos = out['xq'] is not None
        grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        e_op = weight.view_as(weight) if (grad and weight.requires_grad) else prepared.weight
        enc['distance'] = LazyDistance(self._distance, rows if grad else (out['xn'] if normalize else out['x']), e_op,
                                       xq=out['xq'] if cos else (out['xn'] if normalize else out['x']),
                                       eq=prepared.exact_rows() if cos else prepared.weight, metric=ops.metric_name(prepared.metric))
        if out['hist'] is not None:
            enc['hist'] = out['hist']
        memo['encode'] = enc
        memo.update(x=rows, quant=out['idx'])
        memo['decode'] = get_memo(memo, 'decode')
        return z_ste, self._loss_values(memo, m_cb, m_cm, m_vqgan, betas, x), memo

    def _forward_vqkd(self, x: torch.Tensor, memo: Memo):
        cb = self._callbacks.callbacks[0]
        xn, quant, z_ste, value = cb.fused_forward(x, memo)
        memo.update(x=xn, quant=quant)
        memo['decode'] = get_memo(memo, 'decode')
        loss_memo = get_memo(memo, 'loss')
        loss_memo.update({next(iter(self._losses.keys())): value})
        memo['loss'] = loss_memo
        return z_ste, value, memo

    def forward(self, x: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        """quantizers.py:110-117 (BaseQuantizer.forward, then ste(z, memo['x'])).  When nothing customises
        decode/loss the gather, the STE expression and the MSE sums are one kernel with one fused backward; the two
        callback-driven training configs (CVQ-VAE, VQ-KD) are one library call for the whole forward."""
        step = self._one_call_step(x)
        if step is not None:
            return step(x, memo)
        if not self._fusable() or x.dim() != 2:
            z, loss, memo = super().forward(x, memo)
            z = VF.ste(z, memo['x'])
            return z, loss, memo
        x, quant, memo = self.encode(x, memo)
        memo.update(x=x, quant=quant)
        betas = [loss.beta for loss in self._losses.values() if isinstance(loss, VQGANLoss)]
        z_ste, m_cb, m_cm, m_vqgan = VF.fused_decode_loss(x, self._embedding.weight, quant, betas[0] if betas else 0.0)
        memo['decode'] = get_memo(memo, 'decode')
        return z_ste, self._loss_values(memo, m_cb, m_cm, m_vqgan, betas, x), memo


    # ---- the same three entry points on the NCHW feature map (models/base.py:116-146) --------------------
    def map_fusable(self, x: torch.Tensor) -> bool:
        """True when ``forward_map`` / ``encode_map`` take the route without transposes: an NCHW-contiguous fp32 / bf16
        device map, a D with a proposal image, no callback that rewrites the latents before the encode (NormalizeCallback:
        the normalised rows are a new token-major tensor anyway), and a decode/loss tail that can be fused."""
        from .callbacks import BaseCallback
        if not (x.dim() == 4 and x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)):
            return False
        if x.shape[1] != self.embedding_dim or not ops.coarse_supported(x.shape[1]) or x.data_ptr() % 16 or self._cache_codebook:
            return False
        if type(self)._encode is not VectorQuantizer._encode or not hasattr(self._distance, 'encode_map'):
            return False
        # the map entry points are called directly, not through nn.Module.__call__: a registered hook (the one-shot lazy-init
        # pre-hook of LazyInitWeightsMixin, or anything a user attached) would never run — such a module takes the token route
        if len(self._forward_pre_hooks) > 0 or len(self._forward_hooks) > 0:
            return False
        return all(type(cb).before_encode is BaseCallback.before_encode for cb in self._callbacks.callbacks)

    def encode_map(self, x_map: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        """``encode`` for latents given as the feature map [B, D, H, W]: (x_rows [B*H*W, D], quant [B*H*W], memo).  The
        'b c h w -> (b h w) c' of models/base.py:124,140 happens inside the encode's first kernel; ``x_rows`` — the token
        matrix the callbacks and the rest of the step see — is its by-product (detached)."""
        enc = get_memo(memo, 'encode')
        w = self._embedding.weight.detach()
        hist = None
        if self.training and len(self._callbacks.callbacks) > 0:
            hist = torch.empty(self.codebook_size, dtype=torch.int32, device=x_map.device)
        stash = {}
        quant, x_rows = self._distance.encode_map(x_map, w, hist=hist, stash=stash, zero_hist=True)
        enc['distance'] = LazyDistance(self._distance, x_rows, w, xq=stash.get('xq'), eq=stash.get('eq'), metric=stash.get('metric'))
        if hist is not None:
            enc['hist'] = hist
        memo['encode'] = enc
        return x_rows, self._callbacks.after_encode(x_rows, quant, memo), memo

    def forward_map(self, x_map: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, torch.Tensor, Memo]:
        """``forward`` on the feature map: (z_map [B, D, H, W] NCHW-contiguous, loss, memo) — BaseModel.quantize
        (models/base.py:116-128) without either rearrangement kernel.  Gradients flow to ``x_map`` and the codebook."""
        assert self.map_fusable(x_map) and self._fusable()
        x_rows, quant, memo = self.encode_map(x_map, memo)
        memo.update(x=x_rows, quant=quant)
        betas = [loss.beta for loss in self._losses.values() if isinstance(loss, VQGANLoss)]
        z_map, m_cb, m_cm, m_vqgan = VF.fused_map_decode_loss(x_map, x_rows, self._embedding.weight, quant, betas[0] if betas else 0.0)
        memo['decode'] = get_memo(memo, 'decode')
        loss = self._loss_values(memo, m_cb, m_cm, m_vqgan, betas, x_map)
        return z_map, loss, memo

    def decode_map(self, quant: torch.Tensor, memo: Memo) -> tuple[torch.Tensor, Memo]:
        """``decode`` of an image-shaped index tensor [B, H, W] straight into the map [B, D, H, W] (decode_from_quant,
        image_reconstruction/models.py:97-106); no gradient (the reference decodes tokens under no_grad there)."""
        b, h, w = quant.shape
        z_map, _ = ops.gather_ste_map(None, self._embedding.weight.detach(), quant.reshape(-1), b, h, w)
        return z_map, memo


@VQITQuantizerRegistry.register_()
class VQGANQuantizer(VectorQuantizer):

    def _init_weights(self, config: Config) -> bool:
        if Config(config) == Config(type='vqgan'):
            config = Config(type='uniform_', a=-1.0 / self.codebook_size, b=1.0 / self.codebook_size)
        return super()._init_weights(config)


@VQITQuantizerRegistry.register_()
class VQKDQuantizer(VectorQuantizer):

    def _init_weights(self, config: Config) -> bool:
        return False
