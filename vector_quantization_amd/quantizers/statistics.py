"""QuantStatistics — mirror of vq/algorithms/vq/utils.py:13-52 with the histogram taken from the fused
argmin epilogue when available and the collectives packed (utils.all_reduce_statistics)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from ..utils import all_reduce_statistics, get_world_size


class QuantStatistics:

    def __init__(self, *args, quant: torch.Tensor, codebook_size: int, sync: bool = False,
                 hist: Optional[torch.Tensor] = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._quant = quant
        self._codebook_size = codebook_size
        self._sync = sync and get_world_size() > 1
        self._hist = hist
        self._reduced = None

    def _local(self):
        if self._hist is None:
            self._hist = ops.hist(self._quant, self._codebook_size)
        return self._hist.to(torch.int64), self._quant.numel()

    def _statistics(self):
        """(bin_count int64[K], num_elements) — all-reduced together in one collective when syncing."""
        if self._reduced is None:
            hist, numel = self._local()
            if self._sync:
                hist, numel, _ = all_reduce_statistics(hist, numel)
            self._reduced = (hist, numel)
        return self._reduced

    def bin_count(self) -> torch.Tensor:
        return self._statistics()[0]

    def num_elements(self) -> torch.Tensor:
        return self._quant.new_tensor(self._statistics()[1])

    def frequency(self) -> torch.Tensor:
        bin_count, numel = self._statistics()
        return bin_count / numel
