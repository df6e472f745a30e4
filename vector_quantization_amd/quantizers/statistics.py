"""Code-usage statistics of one batch of tokens (the role of ``QuantStatistics``, vq/algorithms/vq/utils.py:13-52).

What callers get is the reference's: ``bin_count()`` int64 [K], ``num_elements()`` a scalar tensor, ``frequency()`` =
bin_count / num_elements, each summed over the ranks when ``sync`` is set and more than one rank runs.  How it is
produced differs: the histogram usually already exists (the argmin epilogue counts code hits as it writes the tokens:
pass it as ``hist``), and histogram and element count cross the wire together in ONE packed int64 all-reduce
(``utils.all_reduce_statistics``) instead of one collective per quantity."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import ops
from ..utils import all_reduce_statistics, get_world_size


class QuantStatistics:

    def __init__(self, *args, quant: torch.Tensor, codebook_size: int, sync: bool = False,
                 hist: Optional[torch.Tensor] = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._tokens, self._K = quant, codebook_size
        self._across_ranks = bool(sync) and get_world_size() > 1
        self._epilogue_hist = hist                  # int32 [K] from the fused argmin, when the caller has it
        self._totals: Optional[Tuple[torch.Tensor, object]] = None

    def _statistics(self) -> Tuple[torch.Tensor, object]:
        """(histogram int64 [K], element count) — computed once; the count is a python int on one rank and a device
        scalar after the all-reduce (no host synchronisation either way)."""
        if self._totals is None:
            counts = self._epilogue_hist if self._epilogue_hist is not None else ops.hist(self._tokens, self._K)
            counts, n = counts.to(torch.int64), self._tokens.numel()
            if self._across_ranks:
                counts, n, _ = all_reduce_statistics(counts, n)
            self._totals = (counts, n)
        return self._totals

    def bin_count(self) -> torch.Tensor:
        return self._statistics()[0]

    def num_elements(self) -> torch.Tensor:
        n = self._statistics()[1]
        return n if isinstance(n, torch.Tensor) else self._tokens.new_tensor(n)

    def frequency(self) -> torch.Tensor:
        counts, n = self._statistics()
        return counts / n
