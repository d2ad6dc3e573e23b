"""Guided-filter reverse pass, round-6 streaming form (csrc/gf_backward.hip: both eps in one stage-1 launch, 48-column strips walked as
runs of rows) -- reverse of Cell_Decom.decomposition (core/model_fusion_auto.py:517-535) under autograd.  Checked against the oracle's
float64 autograd (the bound is the oracle's own float32 error, as in test_backward_fullsize_gpu.py) and against the round-1 kernels
(PAIF_GF_BWD=v1), workspace by workspace."""
import numpy as np
import pytest
import torch

from paif_amd import _lib, ops, synthetic as S
from paif_amd.ops import _p, _stream

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def maxabs(a, b):
    return float((a - b).abs().max())


# (B, H, W, add): one strip / several strips with a ragged last one, runs that cross strip and image boundaries (rows per run does not
# divide H), the smallest legal size, an odd width, a tall narrow map (long runs)
SHAPES = [(1, 20, 24, False), (2, 33, 41, True), (1, 64, 96, True), (3, 10, 10, False), (1, 70, 130, True), (2, 480, 100, False),
          (1, 37, 85, True)]


def _inputs(B, H, W, use_add, smooth):
    g = torch.Generator().manual_seed(B * 1000 + H * 7 + W)
    if smooth:
        x = torch.from_numpy(S.make_smooth_feature(11 + H, B, 32, H, W)).float()
    else:
        x = torch.randn(B, 32, H, W, generator=g)
    dlf = torch.randn(2, B, 32, H, W, generator=g)
    add = torch.randn(B, 32, H, W, generator=g) if use_add else None
    return x, dlf, add


def _run(mode, monkeypatch, guide, xn, ab, dlf, add):
    """-> (t_my, t_mgy, t_g, dy) of one call through the C-ABI entry with caller-owned workspaces"""
    monkeypatch.setenv("PAIF_GF_BWD", mode)
    L = _lib.load()
    B, H, W, _ = xn.shape
    gstat = torch.zeros(L.paif_guided_filter_fused_workspace_floats(B, H, W), device=xn.device)
    t_my, t_mgy, dy = torch.full_like(xn, 7.0), torch.full_like(xn, 7.0), torch.full_like(xn, 7.0)
    t_g = torch.full((B, H, W, 4), 7.0, device=xn.device)
    _lib.check(L.paif_guided_filter_bwd_input(_p(guide), _p(xn), _p(ab), _p(dlf), 1e-3, 1e-4, _p(add), _p(gstat), _p(t_my), _p(t_mgy),
                                              _p(t_g), _p(dy), B, H, W, _stream()), "guided_filter_bwd")
    torch.cuda.synchronize()
    return t_my, t_mgy, t_g[..., :3], dy


@pytest.mark.parametrize("shape", SHAPES)
def test_streaming_reverse_pass_vs_oracle_autograd_and_round1_kernels(shape, monkeypatch):
    from oracle import paif_oracle as O

    B, H, W, use_add = shape
    x, dlf, add = _inputs(B, H, W, use_add, smooth=True)
    dev = _dev()
    xn = ops.to_nhwc(x.to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    _lf, ab = ops.guided_filter_pair(guide, xn, want_ab=True)
    dlfn = torch.stack([ops.to_nhwc(d.to(dev)) for d in dlf]).contiguous()
    addn = ops.to_nhwc(add.to(dev)).contiguous() if use_add else None
    v1 = _run("v1", monkeypatch, guide, xn, ab, dlfn, addn)
    v2 = _run("v2", monkeypatch, guide, xn, ab, dlfn, addn)

    def oracle(dtype):
        xx = x.to(dtype).requires_grad_(True)
        res = O.get_residue(xx)
        loss = sum((O.guided_filter(res, xx, 4, eps) * d.to(dtype)).sum() for eps, d in zip((0.001, 0.0001), dlf))
        loss.backward()
        return xx.grad + (add.to(dtype) if use_add else 0)

    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    floor = maxabs(g32.double(), g64)
    scale = float(g64.abs().max())
    for name, out in (("v1", v1), ("v2", v2)):
        dx = out[3].permute(0, 3, 1, 2).cpu().double()
        assert torch.isfinite(dx).all()
        err = maxabs(dx, g64)
        assert err <= max(1.5 * floor, 1e-4 * scale), (name, shape, err, floor, scale)
    # every workspace row the new kernels write is the round-1 kernels' (nothing stale: the buffers start at 7.0); the two forms
    # differ by the guide statistics' rounding (var = E[g^2] - E[g]^2 is formed by different sums), amplified by 1 / (var + 1e-4)
    for name, a, b in zip(("t_my", "t_mgy", "t_g", "dy"), v1, v2):
        assert maxabs(a, b) <= 2e-3 * max(float(a.abs().max()), 1e-6), (name, shape, maxabs(a, b), float(a.abs().max()))


def test_streaming_reverse_pass_is_what_ops_runs(monkeypatch):
    """ops.guided_filter_bwd (the entry the taped networks call) without the A/B switch = the streaming form, bit for bit."""
    B, H, W = 2, 40, 56
    x, dlf, add = _inputs(B, H, W, True, smooth=False)
    dev = _dev()
    xn = ops.to_nhwc(x.to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    _lf, ab = ops.guided_filter_pair(guide, xn, want_ab=True)
    dlfn = torch.stack([ops.to_nhwc(d.to(dev)) for d in dlf]).contiguous()
    addn = ops.to_nhwc(add.to(dev)).contiguous()
    monkeypatch.delenv("PAIF_GF_BWD", raising=False)
    a = ops.guided_filter_bwd(guide, xn, ab, dlfn, add=addn).clone()
    b = _run("v2", monkeypatch, guide, xn, ab, dlfn, addn)[3]
    assert torch.equal(a, b)
    monkeypatch.setenv("PAIF_GF_BWD", "v3")
    with pytest.raises(RuntimeError, match="PAIF_GF_BWD"):
        ops.guided_filter_bwd(guide, xn, ab, dlfn, add=addn)
