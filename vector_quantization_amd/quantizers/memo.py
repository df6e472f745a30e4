"""Memo plumbing shared by the quantizer modules.

A memo is the plain dict the reference threads through every call (todd.runners.Memo); stages write their by-products
under their own key ('encode', 'decode', 'loss') and callers read what they need (memo side effects).
"""
from __future__ import annotations

from torch import nn

from ..config import Config, RegistryMeta

Memo = dict


def get_memo(memo: Memo, key: str) -> Memo:
    """Sub-memo stored under ``key``; an empty one when the stage has not run yet (vq/utils/misc.py:30-38)."""
    sub = memo.get(key)
    if sub is None:
        return Config()
    assert isinstance(sub, dict), f'memo[{key!r}] is not a memo'
    return sub


class ModuleDict(nn.ModuleDict):
    """Named bundle of modules evaluated together: ``bundle(*args)`` → ``{name: module(*args)}``
    (the behaviour the path needs from todd.patches.torch.ModuleDict)."""

    def forward(self, *args, **kwargs) -> dict:
        out = {}
        for name, module in self.items():
            out[name] = module(*args, **kwargs)
        return out


def build_module_dict(registry: RegistryMeta, config: Config, **kwargs) -> ModuleDict:
    """One module per non-null entry of ``config``, built (or passed through) by ``registry``
    (vq/utils/builders.py:24-34)."""
    built = {}
    for name, entry in config.items():
        if entry is not None:
            built[name] = registry.build_or_return(entry, **kwargs)
    return ModuleDict(built)
