"""Oracle shim for timm.models.layers (un-vendored dependency of core/mix_transformer.py:11).

Written from the package's documented behaviour, NOT from its source: parity unpinned here.
"""
import collections.abc

import torch
import torch.nn as nn

trunc_normal_ = torch.nn.init.trunc_normal_


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return (x, x)


class DropPath(nn.Module):
    """Per-sample stochastic depth: identity in eval or when p == 0."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep
