"""Connectors either side of the quantizer (vq/tasks/image_tokenization/models/connectors/{base,conv}.py) — SURVEY.md
§8f row 3.

The reference runs a 1x1 ``nn.Conv2d`` in NCHW and then rearranges 'b c h w -> (b h w) c' for the quantizer (and back
before ``pre_decode``): two full transposes of the latent map per forward, two more per backward.  On MI355X the
rearrangement is removed rather than accelerated: ``ConvConnector(channels_last=True)`` keeps its weight and its
output in ``torch.channels_last``, where the conv output [B, D, H, W] is byte-for-byte the token matrix [(B H W), D]
the HIP quantizer path consumes (``tokenization.to_tokens`` then returns a view), and the quantizer's output rows are
handed to ``pre_decode`` as a channels-last view as well.  A 1x1 conv in channels-last is one plain library GEMM
[(B H W), C_in] x [C_in, C_out]; it stays with the framework's GEMM library.
"""
from __future__ import annotations

import torch
from torch import nn

from .config import BuildPreHookMixin, Config, Item, RegistryMeta
from .registries import VQITConnectorRegistry


@VQITConnectorRegistry.register_()
class BaseConnector(nn.Module):
    """Identity connector carrying the channel counts (connectors/base.py:12-39)."""

    def __init__(self, *args, in_channels: int, out_channels: int, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._in_channels = in_channels
        self._out_channels = out_channels

    @property
    def in_channels(self) -> int:
        return self._in_channels

    @property
    def out_channels(self) -> int:
        return self._out_channels

    def forward(self, x: torch.Tensor, memo: dict) -> tuple[torch.Tensor, dict]:
        assert x.shape[1] == self._in_channels == self._out_channels
        return x, memo


@VQITConnectorRegistry.register_()
class ConvConnector(BuildPreHookMixin, BaseConnector):
    """Conv connector, 1x1 by default (connectors/conv.py:15-56).  Same constructor, config keys and state-dict
    (``_conv.weight``, ``_conv.bias``); ``channels_last`` only changes the memory format of weight and output."""

    def __init__(self, *args, conv: nn.Conv2d, channels_last: bool = True, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        assert conv.in_channels == self._in_channels
        assert conv.out_channels == self._out_channels
        self._conv = conv
        self._channels_last = channels_last
        if channels_last:
            self._conv.to(memory_format=torch.channels_last)

    @classmethod
    def conv_build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        conv = config.conv if 'conv' in config else Config(kernel_size=1)
        config.conv = nn.Conv2d(config.in_channels, config.out_channels, **conv)
        return config

    @classmethod
    def build_pre_hook(cls, config: Config, registry: RegistryMeta, item: Item) -> Config:
        config = super().build_pre_hook(config, registry, item)
        return cls.conv_build_pre_hook(config, registry, item)

    def forward(self, x: torch.Tensor, memo: dict) -> tuple[torch.Tensor, dict]:
        if self._channels_last:
            x = x.contiguous(memory_format=torch.channels_last)    # no-op when the producer already is channels-last
        return self._conv(x), memo
