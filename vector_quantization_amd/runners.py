"""The runner-level callers of the path, same names and behaviour as the reference's

* ``Tokenizer``            vq/tasks/image_tokenization/runners/tokenizer.py:21-55 — the validator whose iteration is
                           ``model.encode_to_quant(image, memo)``;
* ``TokenizeCallback``     runners/callbacks.py:29-53 — ``tokens/{iter}_{rank}.pth`` = Tokens(id_, category, tokens[b, h, w]);
* ``LlamaGenTokenizeCallback``  tools/tokenize_llamagen.py:65-103 (there also called TokenizeCallback, registered with
                           ``force=True``) — ten-crop codes / labels as ``{i}.npy``;
* ``CodebookUsageMetric`` / ``CodebookPPLMetric``   runners/metrics.py:25-73 — bincount per iteration, one all-reduce at
                           summary time.

The reference builds these through todd's runner machinery (dataset / dataloader / strategy / logging builders, ETA,
checkpoints).  That machinery is control plane and out of scope; what is here is the protocol those four classes live in —
``bind(runner)``, ``before_run_iter`` / ``after_run_iter(batch, memo)``, ``forward(batch, memo)``, ``summary(memo)`` — and a
``Tokenizer.run`` loop that drives it over any iterable of batches.  The histogram and the two summaries run on the device
(``vqhip_hist``, ``vqhip_codebook_metrics``); the encoder in front of the quantizer is whatever callable the model holds.
"""
from __future__ import annotations

import pathlib
import re
import types
from typing import Any, Callable, Iterable, Mapping, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import ops, tokenization
from .registries import VQITCallbackRegistry, VQITMetricRegistry, VQITRunnerRegistry
from .tokenization import Tokens
from .utils import get_rank, get_world_size

__all__ = ['Tokenizer', 'TokenizerModel', 'TokenizeCallback', 'LlamaGenTokenizeCallback', 'Tokens', 'CodebookMixin',
           'CodebookUsageMetric', 'CodebookPPLMetric', 'BaseCallback', 'BaseMetric', 'get_']

_STEP = re.compile(r'''\[\s*(?:"([^"]*)"|'([^']*)'|(-?\d+))\s*\]|\.([A-Za-z_]\w*)''')


def get_(obj: Any, attr: str) -> Any:
    """todd's accessor strings as the configs write them — ``'["quantizer"]["quant"]'`` (configs/vqgan/runner.py:123),
    ``'["ir"]["encode_to_quant"]["quantizer"]["quant"]'`` (configs/ar/runner.py:116) — walked without ``eval``."""
    pos = 0
    for m in _STEP.finditer(attr):
        if m.start() != pos:
            break
        pos = m.end()
        dq, sq, num, name = m.groups()
        if name is not None:
            obj = getattr(obj, name)
        elif num is not None:
            obj = obj[int(num)]
        else:
            obj = obj[dq if dq is not None else sq]
    if pos != len(attr):
        raise ValueError(f'unsupported accessor {attr!r}')
    return obj


class _RunnerHolder:
    def __init__(self, *args, **kwargs) -> None:
        super().__init__()
        self._runner = None

    def bind(self, runner) -> None:
        self._runner = runner

    @property
    def runner(self):
        assert self._runner is not None, f'{type(self).__name__} is not bound to a runner'
        return self._runner


class BaseCallback(_RunnerHolder):
    def before_run_iter(self, batch: Mapping, memo: dict) -> None:
        pass

    def after_run_iter(self, batch: Mapping, memo: dict) -> None:
        pass


class BaseMetric(_RunnerHolder):
    def forward(self, batch: Mapping, memo: dict) -> dict:
        raise NotImplementedError

    def summary(self, memo: dict) -> float:
        raise NotImplementedError


# ---- token files ------------------------------------------------------------------------------------------------------

@VQITCallbackRegistry.register_()
class TokenizeCallback(BaseCallback):
    """runners/callbacks.py:29-53."""

    @property
    def token_dir(self) -> pathlib.Path:
        return pathlib.Path(self.runner.work_dir) / 'tokens'

    def bind(self, *args, **kwargs) -> None:
        super().bind(*args, **kwargs)
        self.token_dir.mkdir(parents=True, exist_ok=True)

    def after_run_iter(self, batch: Mapping, memo: dict) -> None:
        super().after_run_iter(batch, memo)
        quantizer_memo = memo['quantizer']
        tokenization.save_tokens(self.runner.work_dir, self.runner.iter_, batch['id_'], batch['category'],
                                 quantizer_memo['quant'], quantizer_memo['x_shape'])


@VQITCallbackRegistry.register_()
class LlamaGenTokenizeCallback(BaseCallback):
    """tools/tokenize_llamagen.py:65-103: batches are ONE image of ten crops (``[1, 10, C, H, W]``, squeezed before the
    iteration); ``llamagen_tokens/imagenet{S}_codes/{i}.npy`` int64 ``[1, 10, h*w]`` and ``…_labels/{i}.npy``,
    ``i = (iter - 1) * world_size + rank``."""

    @property
    def token_dir(self) -> pathlib.Path:
        return pathlib.Path(self.runner.work_dir) / 'llamagen_tokens'

    @property
    def code_dir(self) -> pathlib.Path:
        return self.token_dir / f'imagenet{self.runner.dataset.image_size}_codes'

    @property
    def label_dir(self) -> pathlib.Path:
        return self.token_dir / f'imagenet{self.runner.dataset.image_size}_labels'

    def bind(self, *args, **kwargs) -> None:
        super().bind(*args, **kwargs)
        self.code_dir.mkdir(parents=True, exist_ok=True)
        self.label_dir.mkdir(parents=True, exist_ok=True)

    def before_run_iter(self, batch, memo: dict) -> None:
        batch['original_image'] = batch['original_image'].squeeze(0)
        batch['image'] = batch['image'].squeeze(0)
        return super().before_run_iter(batch, memo)

    def after_run_iter(self, batch: Mapping, memo: dict) -> None:
        tokenization.save_llamagen(self.runner.work_dir, self.runner.dataset.image_size, self.runner.iter_,
                                   memo['quantizer']['quant'], batch['category'])


# ---- codebook metrics ---------------------------------------------------------------------------------------------------

class CodebookMixin(BaseMetric):
    """runners/metrics.py:25-56.  ``forward`` counts the tokens the accessor points at with one ``vqhip_hist`` launch
    (block-private LDS bins; int64 running totals like the reference's ``bincount``).  Device tokens only: like every
    other entry of this package there is no CPU path (``ops`` raises on a CPU tensor) — the reference's metric also runs
    on CPU tokens.  The summaries are evaluated in float64 on the device (``vqhip_codebook_metrics``) where the reference
    goes through fp32 ``Categorical.entropy``: equal to ~1e-6 relative, not bitwise (tests/test_runners.py)."""

    def __init__(self, *args, quant: str, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._quant = quant
        self._counts: Any = 0

    def bind(self, *args, **kwargs) -> None:
        super().bind(*args, **kwargs)
        self._codebook_size = self.runner.strategy.module.quantizer.codebook_size

    def forward(self, batch: Mapping, memo: dict) -> dict:
        quant: torch.Tensor = get_(memo, self._quant).flatten()
        counts = ops.hist(quant, self._codebook_size).to(torch.int64)
        self._counts = self._counts + counts
        return memo

    def _summary(self, memo: dict, counts: torch.Tensor) -> float:
        raise NotImplementedError

    def summary(self, memo: dict) -> float:
        if isinstance(self._counts, int):
            return 0.
        counts = self._counts.clone()
        if get_world_size() > 1:
            dist.all_reduce(counts)
        return self._summary(memo, counts)


@VQITMetricRegistry.register_()
class CodebookUsageMetric(CodebookMixin):
    """runners/metrics.py:59-63: codes used / K."""

    def _summary(self, memo: dict, counts: torch.Tensor) -> float:
        return float(ops.codebook_metrics(counts)[0].item())


@VQITMetricRegistry.register_()
class CodebookPPLMetric(CodebookMixin):
    """runners/metrics.py:66-73: entropy (nats) of the code histogram — the number docs/pretrained_models.md quotes as PPL."""

    def _summary(self, memo: dict, counts: torch.Tensor) -> float:
        return float(ops.codebook_metrics(counts)[1].item())


# ---- the model the runner drives, and the runner ----------------------------------------------------------------------------

class TokenizerModel(nn.Module):
    """What ``Tokenizer._run_iter`` needs of the reference's BaseModel (models/base.py:29-146): ``encode_to_quant`` and the
    ``quantizer`` property.  ``encoder`` is any callable image -> latent map ``[b, c, h, w]`` (the reference's encoder +
    post_encode connector; identity when the batches already hold latents)."""

    def __init__(self, quantizer: nn.Module, encoder: Optional[Callable[[torch.Tensor], torch.Tensor]] = None) -> None:
        super().__init__()
        self._quantizer = quantizer
        self._encoder = encoder

    @property
    def quantizer(self) -> nn.Module:
        return self._quantizer

    def encode(self, image: torch.Tensor, memo: dict):
        return (image if self._encoder is None else self._encoder(image)), memo

    def encode_to_quant(self, image: torch.Tensor, memo: dict):
        x, memo = self.encode(image, memo)
        return tokenization.encode_to_quant(self._quantizer, x, memo)


@VQITRunnerRegistry.register_()
class Tokenizer:
    """runners/tokenizer.py:21-55.  ``run()`` is the validator loop reduced to what the four classes above observe:
    ``iter_`` counts from 1, callbacks see ``before_run_iter`` / ``after_run_iter``, metrics see ``forward`` per batch and
    ``summary`` once."""

    def __init__(self, *args, model: nn.Module, dataloader: Iterable[Mapping], work_dir, callbacks: Iterable[BaseCallback] = (),
                 metrics: Optional[Mapping[str, BaseMetric]] = None, dataset: Any = None, **kwargs) -> None:
        self.strategy = types.SimpleNamespace(module=model)
        self.dataloader = dataloader
        self.dataset = dataset if dataset is not None else getattr(dataloader, 'dataset', None)
        self.work_dir = pathlib.Path(work_dir)
        self.iter_ = 0
        self.callbacks = list(callbacks)
        self.metrics = dict(metrics or {})
        for holder in (*self.callbacks, *self.metrics.values()):
            holder.bind(self)

    def _run_iter(self, batch: Mapping, memo: dict, *args, **kwargs) -> dict:
        model = self.strategy.module
        original_image = batch['original_image']
        image = batch['image']
        if torch.cuda.is_available():
            original_image = original_image.cuda()
            image = image.cuda()
        memo.update(original_image=original_image, image=image)
        _, memo = model.encode_to_quant(image, memo)
        return memo

    @torch.no_grad()
    def run(self) -> dict:
        memo: dict = {}
        for batch in self.dataloader:
            self.iter_ += 1
            memo = {}
            for c in self.callbacks:
                c.before_run_iter(batch, memo)
            memo = self._run_iter(batch, memo)
            for m in self.metrics.values():
                memo = m.forward(batch, memo)
            for c in self.callbacks:
                c.after_run_iter(batch, memo)
        return {name: m.summary(memo) for name, m in self.metrics.items()}
