"""Guided filter, taped forward and reverse pass in the round-6 streaming form (csrc/gf_taped.hip, csrc/gf_backward.hip: both eps in
one launch, 48-column strips walked as runs of rows, the two-map tape (mean_y, cov)) -- Cell_Decom.decomposition
(core/model_fusion_auto.py:517-535) under autograd.  Checked against the oracle's float64 forward / autograd (the bound is the oracle's
own float32 error, as in test_backward_fullsize_gpu.py) and against the round-1 kernels (PAIF_GF_BWD=v1), workspace by workspace."""
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
    """-> (t_my, t_mgy, t_g, dy) of one call through the C-ABI entries with caller-owned workspaces.  mode: "v1" / "v2" = the round-1 /
    the streaming kernels over the round-1 tape ab; "mc" = the streaming kernels over the streaming forward's tape (an ops.GfTape)."""
    L = _lib.load()
    B, H, W, _ = xn.shape
    t_my, t_mgy, dy = torch.full_like(xn, 7.0), torch.full_like(xn, 7.0), torch.full_like(xn, 7.0)
    t_g = torch.full((B, H, W, 4), 7.0, device=xn.device)
    if mode == "mc":
        _lib.check(L.paif_guided_filter_bwd_input_mc(_p(guide), _p(xn), _p(ab.mc), _p(ab.stats), _p(dlf), _p(add), _p(t_my), _p(t_mgy), _p(t_g),
                                                     _p(dy), B, H, W, _stream()), "guided_filter_bwd_mc")
    else:
        monkeypatch.setenv("PAIF_GF_BWD", mode)
        gstat = torch.zeros(L.paif_guided_filter_fused_workspace_floats(B, H, W), device=xn.device)
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
    lf_ab, ab = ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")
    lf_mc, mc = ops.guided_filter_pair(guide, xn, want_ab=True, tape="mc")
    assert isinstance(mc, ops.GfTape) and tuple(mc.mc.shape) == (2, B, H, W, 32) and tuple(ab.shape) == (4, B, H, W, 32)
    dlfn = torch.stack([ops.to_nhwc(d.to(dev)) for d in dlf]).contiguous()
    addn = ops.to_nhwc(add.to(dev)).contiguous() if use_add else None
    v1 = _run("v1", monkeypatch, guide, xn, ab, dlfn, addn)
    v2 = _run("v2", monkeypatch, guide, xn, ab, dlfn, addn)
    vm = _run("mc", monkeypatch, guide, xn, mc, dlfn, addn)

    def oracle(dtype):
        xx = x.to(dtype).requires_grad_(True)
        res = O.get_residue(xx)
        lfs = [O.guided_filter(res, xx, 4, eps) for eps in (0.001, 0.0001)]
        loss = sum((l * d.to(dtype)).sum() for l, d in zip(lfs, dlf))
        loss.backward()
        return [l.detach() for l in lfs], xx.grad + (add.to(dtype) if use_add else 0)

    (l64, g64), (l32, g32) = oracle(torch.float64), oracle(torch.float32)
    # forward: the streaming pair's LF maps against the oracle's float64 ones, as close as the oracle's own float32 run (the reference's
    # cumsum box filter carries its own noise: 2e-4 abs in test_fusion_gpu.test_guided_filter) -- and next to the round-1 pair's
    for e in range(2):
        ffloor = maxabs(l32[e].double(), l64[e])
        fscale = float(l64[e].abs().max())
        for name, lf in (("ab", lf_ab), ("mc", lf_mc)):
            err = maxabs(lf[e].permute(0, 3, 1, 2).cpu().double(), l64[e])
            assert err <= max(1.5 * ffloor, 2e-6 * fscale), (name, e, shape, err, ffloor, fscale)
    # the tape itself: A_e = cov * rd_e and b_e = mean_y - A_e mean_g re-formed from (mean_y, cov) are the round-1 tape's maps
    npix = B * H * W
    mg, rd0, rd1 = (mc.stats[i * npix:(i + 1) * npix].view(B, H, W, 1) for i in range(3))
    for e, rd in enumerate((rd0, rd1)):
        A = mc.mc[1] * rd
        assert maxabs(A, ab[2 * e]) <= 2e-3 * max(float(ab[2 * e].abs().max()), 1e-6), (shape, e)
        assert maxabs(mc.mc[0] - A * mg, ab[2 * e + 1]) <= 2e-3 * max(float(ab[2 * e + 1].abs().max()), 1e-6), (shape, e)
    # reverse
    floor = maxabs(g32.double(), g64)
    scale = float(g64.abs().max())
    for name, out in (("v1", v1), ("v2", v2), ("mc", vm)):
        dx = out[3].permute(0, 3, 1, 2).cpu().double()
        assert torch.isfinite(dx).all()
        err = maxabs(dx, g64)
        assert err <= max(1.5 * floor, 1e-4 * scale), (name, shape, err, floor, scale)
    # every workspace row the new kernels write is the round-1 kernels' (nothing stale: the buffers start at 7.0); the forms differ by
    # the guide statistics' rounding (var = E[g^2] - E[g]^2 is formed by different sums), amplified by 1 / (var + 1e-4)
    for other in (v2, vm):
        for name, a, b in zip(("t_my", "t_mgy", "t_g", "dy"), v1, other):
            assert maxabs(a, b) <= 2e-3 * max(float(a.abs().max()), 1e-6), (name, shape, maxabs(a, b), float(a.abs().max()))


def test_streaming_pair_is_what_the_taped_networks_run(monkeypatch):
    """ops.guided_filter_pair(want_ab=True) / ops.guided_filter_bwd (the entries the taped networks call) without the A/B switch = the
    streaming forward, its two-map tape and the streaming reverse pass over it, bit for bit; PAIF_GF_BWD=v1 = the round-1 kernels."""
    B, H, W = 2, 40, 56
    x, dlf, add = _inputs(B, H, W, True, smooth=False)
    dev = _dev()
    xn = ops.to_nhwc(x.to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    dlfn = torch.stack([ops.to_nhwc(d.to(dev)) for d in dlf]).contiguous()
    addn = ops.to_nhwc(add.to(dev)).contiguous()
    monkeypatch.delenv("PAIF_GF_BWD", raising=False)
    lf, tape = ops.guided_filter_pair(guide, xn, want_ab=True)
    assert isinstance(tape, ops.GfTape)
    lf_mc, tape_mc = ops.guided_filter_pair(guide, xn, want_ab=True, tape="mc")
    assert torch.equal(lf, lf_mc) and torch.equal(tape.mc, tape_mc.mc)
    a = ops.guided_filter_bwd(guide, xn, tape, dlfn, add=addn).clone()
    assert torch.equal(a, _run("mc", monkeypatch, guide, xn, tape, dlfn, addn)[3])
    monkeypatch.setenv("PAIF_GF_BWD", "v1")
    lf1, tape1 = ops.guided_filter_pair(guide, xn, want_ab=True)
    assert torch.is_tensor(tape1) and tuple(tape1.shape) == (4, B, H, W, 32)
    lf_ab, tape_ab = ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")
    assert torch.equal(lf1, lf_ab) and torch.equal(tape1, tape_ab)
    b = ops.guided_filter_bwd(guide, xn, tape1, dlfn, add=addn).clone()
    assert torch.equal(b, _run("v1", monkeypatch, guide, xn, tape1, dlfn, addn)[3])
    assert maxabs(a, b) <= 2e-3 * float(b.abs().max())
    monkeypatch.setenv("PAIF_GF_BWD", "v3")
    with pytest.raises(RuntimeError, match="PAIF_GF_BWD"):
        ops.guided_filter_bwd(guide, xn, tape1, dlfn, add=addn)


def test_streaming_kernels_do_not_depend_on_the_batch_position(monkeypatch):
    """Three copies of one sample against the B = 1 run, bit for bit: the runs of rows the strips are cut into start at other rows when
    the batch grows, and the 9-row window sums must not care where a run starts (gf_stream.h vsum9: always oldest row first)."""
    H, W = 70, 50
    x, dlf, add = _inputs(1, H, W, True, smooth=True)
    dev = _dev()
    monkeypatch.delenv("PAIF_GF_BWD", raising=False)
    outs = []
    for B in (1, 3):
        xn = ops.to_nhwc(x.repeat(B, 1, 1, 1).to(dev)).contiguous()
        guide = ops.channel_residue(xn)
        lf, tape = ops.guided_filter_pair(guide, xn, want_ab=True)
        dlfn = torch.stack([ops.to_nhwc(d.repeat(B, 1, 1, 1).to(dev)) for d in dlf]).contiguous()
        addn = ops.to_nhwc(add.repeat(B, 1, 1, 1).to(dev)).contiguous()
        dy = ops.guided_filter_bwd(guide, xn, tape, dlfn, add=addn)
        outs.append((lf, tape.mc, dy))
    (lf1, mc1, dy1), (lf3, mc3, dy3) = outs
    for b in range(3):
        assert torch.equal(lf3[:, b], lf1[:, 0]) and torch.equal(mc3[:, b], mc1[:, 0]) and torch.equal(dy3[b], dy1[0]), b
