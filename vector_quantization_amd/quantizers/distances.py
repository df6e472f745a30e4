"""Distances — mirror of vq/algorithms/vq/distances.py:19-46.

``forward(x, e)`` materialises d[N, K] like the reference (used only by consumers that need the matrix);
``argmin(x, e)`` is the fused hot path that never forms it."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch
from torch import nn

from .. import ops
from ..registries import VQITQuantizerDistanceRegistry


class LazyDistance:
    """Stands in for memo['distance'] (vq/algorithms/vq/quantizers.py:98): the [N, K] matrix is produced by the HIP
    distance kernel only if a consumer asks for it."""

    def __init__(self, distance: 'BaseDistance', x: torch.Tensor, e: torch.Tensor) -> None:
        self._distance, self._x, self._e = distance, x, e
        self._value: Optional[torch.Tensor] = None

    @property
    def operands(self):
        return self._x, self._e

    @property
    def metric(self) -> str:
        return self._distance.metric

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self._value = self._distance(self._x, self._e)
        return self._value

    def argmin(self, dim: int) -> torch.Tensor:
        assert dim in (0, -1, 1)
        if dim == 0:        # NearestAnchor: d.argmin(0), fused (never materialises)
            xq, eq = self._distance.exact_operands(self._x, self._e)
            return ops.col_argmin(xq, eq, self.metric)
        return self._distance.argmin(self._x, self._e)


def as_distance_tensor(d) -> torch.Tensor:
    return d.materialize() if isinstance(d, LazyDistance) else d


class BaseDistance(nn.Module, ABC):
    metric: str

    @abstractmethod
    def forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        pass

    def exact_operands(self, x: torch.Tensor, e: torch.Tensor):
        """Operands of the fp32 definition (normalised for cosine)."""
        return x.detach(), e.detach()

    def prepare(self, e: torch.Tensor) -> ops.PreparedCodebook:
        return ops.prepare_codebook(e, self.metric)

    def argmin(self, x: torch.Tensor, e: torch.Tensor, hist: Optional[torch.Tensor] = None,
               prepared: Optional[ops.PreparedCodebook] = None) -> torch.Tensor:
        """torch.argmin(self(x, e), -1) without materialising the matrix."""
        cb = prepared if prepared is not None else self.prepare(e)
        return ops.argmin(x.detach(), cb, hist=hist)


@VQITQuantizerDistanceRegistry.register_()
class L2Distance(BaseDistance):
    metric = 'L2'

    def forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        """torch.cdist(x, e) (mm path), fp32."""
        return ops.distance(x.detach(), e.detach(), 'L2')


@VQITQuantizerDistanceRegistry.register_()
class CosineDistance(BaseDistance):
    metric = 'Cosine'

    def exact_operands(self, x: torch.Tensor, e: torch.Tensor):
        return ops.normalize_rows(x.detach()), ops.normalize_rows(e.detach())

    @staticmethod
    def cosine_similarity(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        """normalize(x) @ normalize(e).T, returned as 1 - distance (the distance kernel's own value)."""
        return 1 - ops.distance(ops.normalize_rows(x.detach()), ops.normalize_rows(e.detach()), 'Cosine')

    def forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        return ops.distance(ops.normalize_rows(x.detach()), ops.normalize_rows(e.detach()), 'Cosine')

    def argmin(self, x, e, hist=None, prepared=None):
        cb = prepared if prepared is not None else self.prepare(e)
        return ops.argmin(ops.normalize_rows(x.detach()), cb, hist=hist)   # the image holds normalize(e)
