"""Oracle shim for guided_filter_pytorch.guided_filter (PyPI guided-filter-pytorch,
wuhuikai/DeepGuidedFilter; version unpinned by the reference; call sites
core/model_fusion_auto.py:2,529-530).

Restates the published algorithm (He et al. guided filter; box filter as cumsum then
difference, border-clipped window) -- parity unpinned: the real package is not available here.
"""
import torch
import torch.nn as nn


def diff_x(t, r):
    left = t[:, :, r:2 * r + 1]
    middle = t[:, :, 2 * r + 1:] - t[:, :, :-2 * r - 1]
    right = t[:, :, -1:] - t[:, :, -2 * r - 1:-r - 1]
    return torch.cat([left, middle, right], dim=2)


def diff_y(t, r):
    left = t[:, :, :, r:2 * r + 1]
    middle = t[:, :, :, 2 * r + 1:] - t[:, :, :, :-2 * r - 1]
    right = t[:, :, :, -1:] - t[:, :, :, -2 * r - 1:-r - 1]
    return torch.cat([left, middle, right], dim=3)


class BoxFilter(nn.Module):
    def __init__(self, r):
        super().__init__()
        self.r = r

    def forward(self, x):
        assert x.dim() == 4
        return diff_y(diff_x(x.cumsum(dim=2), self.r).cumsum(dim=3), self.r)


class GuidedFilter(nn.Module):
    def __init__(self, r, eps=1e-8):
        super().__init__()
        self.r = r
        self.eps = eps
        self.boxfilter = BoxFilter(r)

    def forward(self, x, y):
        n_x, c_x, h_x, w_x = x.size()
        n_y, c_y, h_y, w_y = y.size()
        assert n_x == n_y
        assert c_x == 1 or c_x == c_y
        assert h_x == h_y and w_x == w_y
        assert h_x > 2 * self.r + 1 and w_x > 2 * self.r + 1
        N = self.boxfilter(x.new_ones((1, 1, h_x, w_x)))
        mean_x = self.boxfilter(x) / N
        mean_y = self.boxfilter(y) / N
        cov_xy = self.boxfilter(x * y) / N - mean_x * mean_y
        var_x = self.boxfilter(x * x) / N - mean_x * mean_x
        A = cov_xy / (var_x + self.eps)
        b = mean_y - A * mean_x
        mean_A = self.boxfilter(A) / N
        mean_b = self.boxfilter(b) / N
        return mean_A * x + mean_b
