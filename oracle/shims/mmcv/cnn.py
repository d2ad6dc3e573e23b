"""Oracle shim for mmcv.cnn (un-vendored dependency of core/segformer_head.py:11,50-55).

ConvModule with a norm_cfg: conv bias auto-disabled, order conv -> BN -> ReLU, sub-module
names conv / bn / activate.  Written from documented behaviour: parity unpinned here.
"""
import torch.nn as nn


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, **kw):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=(norm_cfg is None))
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            self.bn = nn.BatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        return self.activate(x)


class DepthwiseSeparableConvModule(nn.Module):  # imported by the reference, never used
    pass
