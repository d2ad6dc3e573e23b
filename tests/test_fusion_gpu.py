"""GPU parity tests (through the C ABI): fusion-network kernels vs the golden vectors captured from the
reference and vs the CPU oracle on the same seeded inputs.

Tolerances (fp32 kernels): max-abs <= 1e-4 on fused / O(1) activations relative to the
tensor's scale; guided-filter LF <= 2e-4 (the reference's own fp32-vs-fp64 floor on LF is ~1e-4: its
cumsum box filter is LESS accurate than the direct window sums used here)."""
import numpy as np
import pytest
import torch

from paif_amd import synthetic as S
from tests import helpers as Hh
from tests.helpers import t, maxabs

pytestmark = pytest.mark.gpu

HIP_PRIMS = list(Hh.PRIMITIVES)   # all 12 primitive strings of the search space ([probe] list)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _scale(a):
    return max(1.0, float(np.abs(a).max()))


@pytest.mark.parametrize("prim", HIP_PRIMS)
def test_primitive_forward(golden, prim):
    from paif_amd.core.model_fusion_auto import MixedOp

    g = golden("ga_primitives")
    op = MixedOp(32, prim).eval()
    S.load_formula_weights(op, salt=Hh.PRIMITIVES.index(prim) + 1)
    op = op.to(_dev())
    x = t(S.make_smooth_feature(11, 1, 32, 24, 32)).to(_dev())
    with torch.no_grad():
        y = op(x)
    assert tuple(y.shape) == (1, 32, 24, 32)
    assert maxabs(y.cpu(), g[prim + ".y"]) <= 1e-4 * _scale(g[prim + ".y"])


def test_primitive_vs_oracle_ragged_size():
    """Edge tiles: H, W not multiples of the 8x32 conv tile; compared with the CPU oracle."""
    from oracle import paif_oracle as O
    from paif_amd.core.model_fusion_auto import MixedOp
    from tests.test_oracle_golden import op_sd

    x = t(S.make_smooth_feature(77, 2, 32, 21, 45))
    for prim in ("Denseblocks_3_1", "Residualblocks_7_1", "DilConv_3_2", "ECAattention_3"):
        op = MixedOp(32, prim).eval()
        S.load_formula_weights(op, salt=Hh.PRIMITIVES.index(prim) + 1)
        with torch.no_grad():
            y = op.to(_dev())(x.to(_dev())).cpu()
            ref = O.mixed_op(x, op_sd(prim), "", prim)
        assert maxabs(y, ref) <= 1e-4 * _scale(ref.numpy()), prim


def test_guided_filter(golden):
    from paif_amd import ops

    g = golden("gb_guided_filter")
    y = t(S.make_smooth_feature(21, 1, 32, 24, 32)).to(_dev())
    ynhwc = ops.to_nhwc(y)
    guide = ops.channel_residue(ynhwc)
    lf = ops.guided_filter_pair(guide, ynhwc)
    for i, eps in enumerate((1e-3, 1e-4)):
        mine = lf[i].permute(0, 3, 1, 2).cpu()
        # vs the reference's fp32 run AND vs its fp64 run (we must be at least as close to fp64 as it is)
        assert maxabs(mine, g["lf_eps%g" % eps]) <= 2e-4
        floor = maxabs(g["lf_eps%g" % eps], g["lf64_eps%g" % eps])
        assert maxabs(mine, g["lf64_eps%g" % eps]) <= max(2.0 * floor, 2e-5)
    with pytest.raises(AssertionError):
        ops.guided_filter_pair(torch.zeros(1, 9, 20, device=_dev()), torch.zeros(1, 9, 20, 32, device=_dev()))


@pytest.mark.parametrize("shape", [(1, 24, 32), (2, 50, 70), (1, 130, 97), (1, 10, 10), (1, 251, 49)])
def test_guided_filter_fused_matches_two_kernel_form(shape):
    """The inference path fuses both stages (coefficient maps stay on chip); the gradient path keeps them.
    Both must agree on ragged sizes (strip / segment / image borders in every combination)."""
    from paif_amd import ops

    B, H, W = shape
    y = ops.to_nhwc(t(S.make_smooth_feature(5, B, 32, H, W)).to(_dev()))
    guide = ops.channel_residue(y)
    fused = ops.guided_filter_pair(guide, y)
    two, _ab = ops.guided_filter_pair(guide, y, want_ab=True)
    assert torch.isfinite(fused).all()
    # 1/n and 1/(var+eps) are formed once and multiplied in the fused kernel (divisions in the two-kernel form):
    # a few ulp through A = cov / (var + eps)
    assert maxabs(fused.cpu(), two.cpu()) <= 1e-5 * max(1.0, float(two.abs().max()))


@pytest.mark.parametrize("kh,dil,nsrc,nres,act", [(3, 1, 1, 0, 0), (3, 1, 2, 1, 1), (3, 1, 3, 3, 0), (3, 2, 1, 0, 2),
                                                 (1, 1, 3, 0, 0), (1, 1, 1, 2, 1), (3, 2, 2, 1, 0)])
def test_dense_conv_persistent_kernel_ragged_edges(kh, dil, nsrc, nres, act):
    """Shapes with >= 1024 tiles take the wave-specialised persistent kernel (conv_mfma.hip).  A ragged image
    (partial tiles on both edges, halo clamps on all four borders, an odd number of stages per workgroup and
    workgroups with no tile at all) is checked against the CPU oracle's conv and against the exact-fp32 MFMA
    kernel, which never takes that path."""
    from paif_amd import ops

    B, H, W = 2, 333, 517      # 42 x 17 x 2 = 1428 tiles
    g = torch.Generator().manual_seed(kh * 100 + dil * 10 + nsrc)
    xs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nsrc)]
    rs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nres)]
    w = torch.randn(32, 32 * nsrc, kh, kh, generator=g) * 0.05
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.1
    slope = torch.tensor([0.2])
    ref = torch.nn.functional.conv2d(torch.cat(xs, 1), w, padding=dil * (kh - 1) // 2, dilation=dil)
    ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act == 1:
        ref = torch.where(ref >= 0, ref, ref * slope)
    elif act == 2:
        ref = ref.clamp_min(0)
    ref = ref * 0.5
    for r in rs:
        ref = ref + r
    dev = _dev()
    xn = [ops.to_nhwc(x.to(dev)) for x in xs]
    rn = [ops.to_nhwc(r.to(dev)) for r in rs]
    outs = {}
    for prec in ("bf16x3", "f32"):
        wpk = ops.pack_conv_weight(w.to(dev), nsrc, 32, kh, precision=prec)
        y = ops.conv2d(xn, wpk, kh, dil=dil, scale=scale.to(dev), shift=shift.to(dev), act=act,
                       prelu=slope.to(dev) if act == 1 else None, alpha=0.5, res=tuple(rn))
        outs[prec] = y.permute(0, 3, 1, 2).cpu()
    sc = float(ref.abs().max())
    assert maxabs(outs["f32"], ref) <= 2e-5 * sc
    assert maxabs(outs["bf16x3"], ref) <= 1e-4 * sc
    assert maxabs(outs["bf16x3"], outs["f32"]) <= 1e-4 * sc


@pytest.mark.parametrize("kh,dil,nsrc,nres,act", [
    (1, 1, 1, 0, 0), (1, 1, 2, 0, 1), (1, 1, 3, 0, 2),                       # 1x1 without residual maps
    (3, 2, 1, 0, 0), (3, 2, 1, 1, 1), (3, 2, 1, 2, 2), (3, 2, 1, 3, 1),      # dilation 2 / one source, residual storer path NR = 0..3
])
def test_dense_conv_persistent_kernel_is_dispatched_and_correct(kh, dil, nsrc, nres, act):
    """The 1x1 configurations the library routes to the persistent wave-specialised kernel (conv_bf16x3_ws) and the dilation-2 3x3
    ones it used to (fp32 maps: now the tile-per-workgroup kernel), at >= 1024 tiles with ragged edges: the dispatch is ASSERTED (kernel tag of the launch) and the result is checked against
    torch.nn.functional.conv2d on the CPU -- an independent reference, not the repo's own fp32 kernel."""
    from paif_amd import ops

    B, H, W = 2, 331, 523      # 42 x 17 x 2 = 1428 tiles of 8 x 32, ragged in both directions
    g = torch.Generator().manual_seed(7000 + kh * 100 + dil * 10 + nsrc + 5 * nres)
    xs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nsrc)]
    rs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nres)]
    w = torch.randn(32, 32 * nsrc, kh, kh, generator=g) * 0.05
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    slope = torch.tensor([0.2])
    ref = torch.nn.functional.conv2d(torch.cat(xs, 1), w, padding=dil * (kh - 1) // 2, dilation=dil)
    ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = torch.where(ref >= 0, ref, ref * slope) if act == 1 else (ref.clamp_min(0) if act == 2 else ref)
    ref = ref * 0.5 + sum(rs) if rs else ref * 0.5
    dev = _dev()
    timer = ops.KernelTimer(lambda tag: True)
    ops.TIMER = timer
    try:
        wpk = ops.pack_conv_weight(w.to(dev), nsrc, 32, kh, precision="bf16x3")
        y = ops.conv2d([ops.to_nhwc(x.to(dev)) for x in xs], wpk, kh, dil=dil, scale=scale.to(dev), shift=shift.to(dev), act=act,
                       prelu=slope.to(dev) if act == 1 else None, alpha=0.5, res=tuple(ops.to_nhwc(r.to(dev)) for r in rs))
    finally:
        ops.TIMER = None
    torch.cuda.synchronize()
    # fp32 maps: 1x1 -> the persistent form; dilation-2 3x3 -> the tile-per-workgroup kernel since round 4 (faster inside the forward;
    # the persistent dilation-2 form serves bf16-stored maps: tests/test_bf16_storage_gpu.py)
    want = "conv_bf16x3_ws<1, 1, 0>" if kh == 1 else "conv_mfma_bf16x3<3, 2, false, 0>"
    assert list(timer.summary()) == [want], list(timer.summary())
    assert maxabs(y.permute(0, 3, 1, 2).cpu(), ref) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("nsrc,nres,act,in_act,pool", [
    (1, 0, 0, 0, False), (1, 1, 1, 0, False), (1, 3, 2, 0, False), (1, 0, 1, 1, False), (1, 2, 0, 2, False),   # resident-B persistent
    (1, 0, 0, 1, True),                                                                                     # pooled: tile-per-workgroup
    (2, 0, 0, 0, False), (2, 2, 1, 0, False), (3, 0, 2, 0, False), (3, 3, 1, 0, False),                     # multi-source
])
def test_dense_conv_3x3_kernel_variants_are_dispatched_and_correct(nsrc, nres, act, in_act, pool):
    """The 3x3 dilation-1 forward conv has three split-bf16 kernels (conv_mfma.hip): the resident-B persistent kernel
    (one source, >= 2048 tiles), the multi-source kernel (2-3 sources) and the tile-per-workgroup kernel (everything else,
    e.g. with the fused ECA pool).  On a ragged shape (H not a multiple of 4 or 8, W not a multiple of 32, 2178 tiles, a
    persistent workgroup count that does not divide them) the dispatch is ASSERTED and the result compared with
    torch.nn.functional.conv2d on the CPU, including input activations, residual maps and the pool partial sums."""
    from paif_amd import ops

    B, H, W = 1, 523, 1051     # 66 x 33 tiles of 8 x 32
    g = torch.Generator().manual_seed(9100 + 100 * nsrc + 10 * nres + act + 3 * in_act)
    xs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nsrc)]
    rs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nres)]
    w = torch.randn(32, 32 * nsrc, 3, 3, generator=g) * 0.05
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    slope, in_slope = torch.tensor([0.2]), torch.tensor([0.3])
    xin = torch.cat(xs, 1)
    xin = torch.where(xin >= 0, xin, xin * in_slope) if in_act == 1 else (xin.clamp_min(0) if in_act == 2 else xin)
    ref = torch.nn.functional.conv2d(xin, w, padding=1)
    ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = torch.where(ref >= 0, ref, ref * slope) if act == 1 else (ref.clamp_min(0) if act == 2 else ref)
    ref = ref * 0.5 + sum(rs) if rs else ref * 0.5
    dev = _dev()
    timer = ops.KernelTimer(lambda tag: True)
    ops.TIMER = timer
    try:
        wpk = ops.pack_conv_weight(w.to(dev), nsrc, 32, 3, precision="bf16x3")
        out = ops.conv2d([ops.to_nhwc(x.to(dev)) for x in xs], wpk, 3, scale=scale.to(dev), shift=shift.to(dev), act=act,
                         prelu=slope.to(dev) if act == 1 else None, alpha=0.5, res=tuple(ops.to_nhwc(r.to(dev)) for r in rs),
                         in_act=in_act, in_prelu=in_slope.to(dev) if in_act == 1 else None, pool=pool)
    finally:
        ops.TIMER = None
    torch.cuda.synchronize()
    y, partial = out if pool else (out, None)
    want = ("conv_mfma_bf16x3<3, 1, false, 0>" if pool else "conv_bf16x3_res<3, 1, 1, 4, 0>") if nsrc == 1 else "conv_bf16x3_ms<3, 1, %d, 0>" % nsrc
    assert list(timer.summary()) == [want], list(timer.summary())
    sc = float(ref.abs().max())
    assert maxabs(y.permute(0, 3, 1, 2).cpu(), ref) <= 1e-4 * sc
    if pool:
        assert maxabs(partial.sum(0).cpu().double(), ref.double().sum((0, 2, 3))) <= 1e-5 * sc * H * W


@pytest.mark.parametrize("nres,act,in_act", [(0, 1, 0), (2, 0, 1)])
def test_dense_conv_resident_kernel_batched_ragged(nres, act, in_act):
    """conv_bf16x3_res on a second shape class: batch 3 (tile ranges of the 8 XCDs straddle image boundaries), H = 258 (2 rows in the
    last 4-row tile), W = 704 (full 32-pixel tiles), against torch on the CPU."""
    from paif_amd import ops

    B, H, W = 3, 258, 704      # 3 x 33 x 22 = 2178 tiles of 8 x 32
    g = torch.Generator().manual_seed(9300 + 10 * nres + act + 3 * in_act)
    x = torch.randn(B, 32, H, W, generator=g)
    rs = [torch.randn(B, 32, H, W, generator=g) for _ in range(nres)]
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.05
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    slope, in_slope = torch.tensor([0.2]), torch.tensor([0.3])
    xin = torch.where(x >= 0, x, x * in_slope) if in_act == 1 else x
    ref = torch.nn.functional.conv2d(xin, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = torch.where(ref >= 0, ref, ref * slope) if act == 1 else ref
    ref = ref * 0.5 + sum(rs) if rs else ref * 0.5
    dev = _dev()
    timer = ops.KernelTimer(lambda tag: True)
    ops.TIMER = timer
    try:
        wpk = ops.pack_conv_weight(w.to(dev), 1, 32, 3, precision="bf16x3")
        y = ops.conv2d([ops.to_nhwc(x.to(dev))], wpk, 3, scale=scale.to(dev), shift=shift.to(dev), act=act,
                       prelu=slope.to(dev) if act == 1 else None, alpha=0.5, res=tuple(ops.to_nhwc(r.to(dev)) for r in rs),
                       in_act=in_act, in_prelu=in_slope.to(dev) if in_act == 1 else None)
    finally:
        ops.TIMER = None
    torch.cuda.synchronize()
    assert list(timer.summary()) == ["conv_bf16x3_res<3, 1, 1, 4, 0>"], list(timer.summary())
    assert maxabs(y.permute(0, 3, 1, 2).cpu(), ref) <= 1e-4 * float(ref.abs().max())


def test_dense_conv_seeded_shape_sweep():
    """40 seeded random configurations (shape, kernel size, dilation, sources, residuals, activation, affine on/off) of the
    dense conv: the split-bf16 kernel the library picks against the exact-fp32 MFMA kernel, and -- on the small shapes --
    against torch.nn.functional.conv2d on the CPU.  Ragged edges, odd sizes, tiny images.  (The persistent kernel's own
    configurations are forced and asserted in test_dense_conv_persistent_kernel_is_dispatched_and_correct; of this sweep's
    seeds only the tile-per-workgroup kernels are guaranteed.)"""
    import random
    from paif_amd import ops

    rnd = random.Random(20261003)
    dev = _dev()
    for it in range(40):
        kh, dil = rnd.choice([(1, 1), (3, 1), (3, 1), (3, 2), (5, 1), (7, 1)])
        nsrc, nres, act = rnd.randint(1, 3), rnd.randint(0, 3), rnd.randint(0, 2)
        if it % 4 == 0:   # large enough for the persistent kernel (>= 1024 tiles of 8 x 32)
            B, H, W = rnd.choice([(1, 8 * rnd.randint(40, 50) + rnd.randint(0, 7), 32 * rnd.randint(28, 34) + rnd.randint(0, 31)),
                                  (3, 131 + rnd.randint(0, 20), 353 + rnd.randint(0, 40))])
        else:
            B, H, W = rnd.randint(1, 3), rnd.randint(1, 70), rnd.randint(1, 90)
        g = torch.Generator().manual_seed(1000 + it)
        xs = [ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)) for _ in range(nsrc)]
        rs = [ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)) for _ in range(nres)]
        w = (torch.randn(32, 32 * nsrc, kh, kh, generator=g) * 0.05).to(dev)
        affine = rnd.random() < 0.7
        scale = (torch.rand(32, generator=g) + 0.5).to(dev) if affine else None
        shift = (torch.randn(32, generator=g) * 0.1).to(dev) if affine else None
        slope = torch.tensor([0.25], device=dev)
        outs = []
        for prec in ("f32", "bf16x3"):
            wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision=prec)
            outs.append(ops.conv2d(xs, wpk, kh, dil=dil, scale=scale, shift=shift, act=act, prelu=slope if act == 1 else None,
                                   alpha=0.75, res=tuple(rs)))
        sc = float(outs[0].abs().max())
        assert torch.isfinite(outs[1]).all()
        assert maxabs(outs[1].cpu(), outs[0].cpu()) <= 1e-4 * max(sc, 1.0), (it, kh, dil, nsrc, nres, act, B, H, W)
        if B * H * W <= 20000:   # independent reference on the CPU
            ref = torch.nn.functional.conv2d(torch.cat([x.permute(0, 3, 1, 2).cpu() for x in xs], 1), w.cpu(),
                                             padding=dil * (kh - 1) // 2, dilation=dil)
            if affine:
                ref = ref * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)
            ref = torch.where(ref >= 0, ref, ref * 0.25) if act == 1 else (ref.clamp_min(0) if act == 2 else ref)
            ref = ref * 0.75
            for r in rs:
                ref = ref + r.permute(0, 3, 1, 2).cpu()
            assert maxabs(outs[0].permute(0, 3, 1, 2).cpu(), ref) <= 3e-5 * max(sc, 1.0), (it, kh, dil, nsrc, nres, act, B, H, W)


def _fusion_net(prefix=""):
    """prefix='enhance_net.' gives the weights the fusion net has INSIDE the composite model's goldens
    (the formula is keyed on the full state_dict key)."""
    from oracle.paif_oracle import FUSION_AT
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched

    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    sd = {k: t(S.formula_tensor(prefix + k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    return net.to(_dev())


def test_fusion_state_dict_layout():
    net = _fusion_net()
    want = {k[len("enhance_net."):]: v for k, v in Hh.layout("mit_b0").items() if k.startswith("enhance_net.")}
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want


def test_fusion_net_intermediates(golden):
    from paif_amd import ops

    g = golden("gc_fusion_48x64")
    net = _fusion_net()
    ir, vis, _ = S.make_batch(1, 48, 64)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    inter = {}
    with torch.no_grad():
        fused = net(t(ir).to(_dev()), ycc[:, 0:1], inter=inter)
    for k in ("fir", "ir_feature", "vis_feature", "feature2"):
        got = inter[k].permute(0, 3, 1, 2).cpu()
        assert maxabs(got, g[k]) <= 1e-4 * _scale(g[k]), k
    assert maxabs(fused.cpu(), g["fused"]) <= 1e-4


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_fusion_forward_at_exact_conv_precision_keeps_the_tail_exact(golden, precision):
    """under set_conv_precision("f32" | "bf16x6") the stem_out tail must run at the requested precision too (the
    one-kernel fp32-map form takes x as bf16 hi + lo: split-bf16 operand precision, 3e-5).  The whole forward is then within
    1e-5 of the reference's fused plane at 48x64 (split-bf16 default: 1e-4 budget), and the fused tail kernel is not launched."""
    from paif_amd import ops

    g = golden("gc_fusion_48x64")
    net = _fusion_net()
    ir, vis, _ = S.make_batch(1, 48, 64)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    calls = []
    orig = ops.stem_out_fused
    ops.stem_out_fused = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    old = ops.CONFIG["conv_precision"]
    try:
        ops.set_conv_precision(precision)
        with torch.no_grad():
            fused = net(t(ir).to(_dev()), ycc[:, 0:1])
        assert not calls, "the split-operand stem_out kernel ran under conv precision %s" % precision
        ops.set_conv_precision("bf16x3")
        with torch.no_grad():
            net(t(ir).to(_dev()), ycc[:, 0:1])
        assert calls, "the default arithmetic takes the one-kernel tail"
    finally:
        ops.set_conv_precision(old)
        ops.stem_out_fused = orig
    assert maxabs(fused.cpu(), g["fused"]) <= 1e-5


def test_showfeatures_forward2(golden):
    """Feature-visualisation path: same parameters as the fusion net (identical state_dict keys),
    forward2 returns the fused image and the decomposition intermediates of the reference."""
    from oracle.paif_oracle import FUSION_AT
    from paif_amd.core import model_fusion_auto as M
    from paif_amd import ops

    g = golden("gj_showfeatures_40x56")
    net = M.Network_Fusion_Searched_showfeatures(32, None, FUSION_AT).eval()
    net.load_state_dict({k: t(S.formula_tensor(k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()}, strict=True)
    assert list(net.state_dict().keys()) == list(_fusion_net().state_dict().keys())
    net = net.to(_dev())
    ir, vis, _ = S.make_batch(1, 40, 56)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    outs = net.forward2(t(ir).to(_dev()), ycc[:, 0:1])
    names = ("fused", "ir_feature", "vis_feature", "lf_ir", "hf_ir", "res_ir", "lf_vis", "hf_vis", "res_vis")
    assert len(outs) == len(names)
    for n, o in zip(names, outs):
        assert tuple(o.shape) == tuple(g[n].shape), n
        # the guided-filter outputs carry the reference's own cumsum noise: 2e-4 abs as in test_guided_filter
        assert maxabs(o.cpu(), g[n]) <= max(2e-4, 1e-4 * _scale(g[n])), n
    # plain forward of the same module = the fused image
    assert maxabs(net(t(ir).to(_dev()), ycc[:, 0:1]).cpu(), g["fused"]) <= 1e-4
    # Cell_Decom_decom.decomposition materialises (LF, HF, res) like the reference's
    lf, hf, res = net.decompation.decomposition(outs[1].contiguous())
    assert lf.shape[1] == 64 and hf.shape[1] == 64 and res.shape[1] == 1
    assert maxabs((lf + hf).cpu(), torch.cat([outs[1], outs[1]], 1).cpu()) <= 1e-5


def test_fused_image_writer_postprocessing(golden, tmp_path):
    """Harness post-processing (test_original.py:181-203): uint8 fused images on the GPU, exact
    against the golden made by executing the reference's own lines; PNG round trip."""
    from paif_amd import harness
    from PIL import Image

    g = golden("gk_fused_writer_2x48x64")
    _, vis, _ = S.make_batch(2, 48, 64)
    img = harness.fused_to_uint8(t(g["fused"]).to(_dev()), t(vis).to(_dev()))
    assert img.dtype == torch.uint8 and tuple(img.shape) == g["fused_image"].shape
    diff = (img.cpu().numpy().astype(np.int32) - g["fused_image"].astype(np.int32))
    # the first quantisation truncates 255*x: an fp32 last-bit difference in the colour transform can move a value
    # across an integer; the reference's own k-ordered fma chain is reproduced, so the result is exact
    assert np.abs(diff).max() == 0, (np.abs(diff).max(), int((diff != 0).sum()))
    harness.write_fused_pngs(img, ["a.png", "b.png"], str(tmp_path))
    back = np.asarray(Image.open(str(tmp_path / "b.png")))
    assert np.array_equal(back, g["fused_image"][1])


def test_fusion_net_b2(golden):
    from paif_amd import ops

    g = golden("gc_fusion_2x64x96")
    net = _fusion_net()
    ir, vis, _ = S.make_batch(2, 64, 96)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    with torch.no_grad():
        fused = net(t(ir).to(_dev()), ycc)  # 3-channel input: forward slices channel 0 itself (:626-627)
    assert maxabs(fused.cpu(), g["fused"]) <= 1e-4


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_two_stream_forward_is_bit_identical(storage):
    """ops.CONFIG["two_stream"]: the infrared and the visible stream of the fusion network on two HIP streams (fork at the stems, join at
    the spatial blend).  Same kernels, same operands: the fused image must be BIT-identical to the single-stream forward -- on repeated
    calls (the side stream's allocator pool is reused across forwards), with another size in between, and inside a captured hipGraph."""
    from paif_amd import ops

    old = dict(ops.CONFIG)
    try:
        ops.set_storage(storage)
        net = _fusion_net()
        outs = {}
        for shape in ((2, 64, 96), (1, 200, 328)):
            ir, vis, _ = S.make_batch(*shape)
            irt = t(ir).to(_dev())
            ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
            with torch.no_grad():
                ops.CONFIG["two_stream"] = False
                ref = net(irt, ycc).clone()
                ops.CONFIG["two_stream"] = True
                for _ in range(4):
                    got = net(irt, ycc)
                    assert torch.equal(got, ref)
            outs[shape] = (irt, ycc, ref)
        irt, ycc, ref = outs[(2, 64, 96)]
        with torch.no_grad():
            assert torch.equal(net(irt, ycc), ref)                 # back to the first size: pool blocks of the other size in between
            gstream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
            with torch.cuda.stream(gstream):
                net(irt, ycc)
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=gstream):
                    gout = net(irt, ycc)
            torch.cuda.synchronize()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(gout, ref)
    finally:
        ops.CONFIG.clear()
        ops.CONFIG.update(old)


def test_colour_glue_and_batch_coupling(golden):
    from paif_amd import ops

    g = golden("gd_colour_glue")
    net = _fusion_net("enhance_net.")
    ir, vis, _ = S.make_batch(2, 64, 96)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    assert maxabs(ycc.cpu(), g["ycc"]) <= 1e-6
    for B, key in ((2, "b2"), (1, "b1")):
        with torch.no_grad():
            fused = net(t(ir[:B]).to(_dev()), ycc[:B].contiguous())
            seg_in = ops.seg_input_from_fused(fused, ycc[:B].contiguous())
        assert maxabs(seg_in.cpu(), g["seg_in_" + key]) <= 5e-4  # |values| up to ~2.6; x255/std amplifies 1e-4 by ~4.4


def test_fusion_full_size_480x640(golden):
    """BASELINE config-2 shape (one pair): fused image against the reference's own output."""
    from paif_amd import ops

    g = golden("gf_model_b3_1x480x640")
    net = _fusion_net("enhance_net.")
    ir, vis, _ = S.make_batch(1, 480, 640)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    with torch.no_grad():
        fused = net(t(ir).to(_dev()), ycc)
    # At full size the reference's OWN fp32 run is ~1e-4 away from its fp64 run (cumsum box filter error
    # grows towards the far image edges).  The HIP path (direct window sums) must be (a) at least as
    # close to the fp64 result as the reference's fp32 run is, and (b) within 2x that floor of the fp32 run.
    floor = maxabs(g["fused"], g["fused64"])
    assert floor < 5e-4
    assert maxabs(fused.cpu(), g["fused64"]) <= max(floor, 2e-5)
    assert maxabs(fused.cpu(), g["fused"]) <= 2.0 * floor + 2e-5


def test_train_mode_forward_uses_batch_statistics():
    """Train-mode forward (BatchNorm batch statistics in DilConv / ResidualModule, running statistics updated) vs the oracle."""
    from oracle import paif_oracle as O

    net = _fusion_net()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    net.train()
    ir, vis, _ = S.make_batch(2, 40, 56)
    ycc = O.rgb2ycrcb(t(vis))
    with torch.no_grad():
        fused = net(t(ir).to(_dev()), ycc[:, 0:1].contiguous().to(_dev()))
    O.TRAIN = O.TrainCtx(0)
    try:
        with torch.no_grad():
            ref = O.fusion_forward(t(ir), ycc[:, 0:1], sd)
    finally:
        O.TRAIN = None
    assert maxabs(fused.cpu(), ref) <= 1e-4
    for k, v in net.state_dict().items():
        if "running_" in k:
            assert maxabs(v.cpu(), sd[k]) <= 1e-5, k
    net.eval()


def test_unknown_and_malformed_primitives():
    from paif_amd.core.model_fusion_auto import MixedOp

    with pytest.raises(KeyError):
        MixedOp(32, "Nope_3_1")
    with pytest.raises(IndexError):
        MixedOp(32, "DilConv_3")


@pytest.mark.parametrize("mengine", ["mfma2"])
@pytest.mark.parametrize("shape", [(1, 24, 32), (2, 64, 96), (1, 50, 131), (3, 37, 49), (2, 11, 10), (1, 100, 47), (1, 130, 200), (1, 480, 640),
                                   (1, 1100, 40), (1, 12, 700), (5, 97, 33)])
def test_guided_filter_matrix_core_engine_vs_valu_engine_and_oracle(shape, mengine):
    """The matrix-core engine -- csrc/gf_mfma2.hip (round 4: two columns per lane, two waves per SIMD; the round-3 engine it superseded
    left the library in round 6, tools/parked/gf_mfma.hip); horizontal box sums as f16 hi/lo band-matrix MFMAs --
    against the all-VALU kernel (PAIF_GF_ENGINE=valu) and the float64 oracle, incl. ragged and odd widths, strips hanging over
    the image edge, several row segments, B > 1, the bench shape, an image taller than one run of rows of the round-4 engine (its
    per-iteration 1 / ny table: pieces of <= 1000 rows) and one wider than ten strips.  Core/model_fusion_auto.py:522-535."""
    import os

    from oracle import paif_oracle as O
    from paif_amd import ops

    B, H, W = shape
    x = t(S.make_smooth_feature(71, B, 32, H, W))
    xn = ops.to_nhwc(x.to(_dev()))
    guide = ops.channel_residue(xn)
    old = os.environ.get("PAIF_GF_ENGINE")
    try:
        os.environ["PAIF_GF_ENGINE"] = "valu"
        a = ops.guided_filter_pair(guide, xn).clone()
        os.environ["PAIF_GF_ENGINE"] = mengine
        b = ops.guided_filter_pair(guide, xn).clone()
    finally:
        if old is None:
            os.environ.pop("PAIF_GF_ENGINE", None)
        else:
            os.environ["PAIF_GF_ENGINE"] = old
    assert maxabs(a, b) <= 5e-6
    if H * W > 100000:
        return                                           # the float64 oracle on the host: small shapes only
    x64 = x.double()
    res = O.get_residue(x64)
    for e, eps in enumerate((0.001, 0.0001)):
        ref = O.guided_filter(res, x64, 4, eps)
        err_m = maxabs(b[e].permute(0, 3, 1, 2).cpu().double(), ref)
        err_v = maxabs(a[e].permute(0, 3, 1, 2).cpu().double(), ref)
        assert err_m <= max(2.0 * err_v, 2e-6), (shape, eps, err_m, err_v)      # as close to float64 as the fp32 direct sums


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 50, 131), (3, 37, 49), (1, 100, 47), (1, 130, 200)])
@pytest.mark.parametrize("engine", ["mfma2", "valu"])
def test_guided_filter_bf16_output_is_the_rounded_fp32_output(shape, engine):
    """The bf16-output form of the fused guided filter (paif_guided_filter_fused_fwd_bf16: the bf16 configuration's storage of the two
    low-frequency maps; channel pairs exchanged by DPP and stored as dwords) computes the same fp32 values and rounds them to nearest
    even -- bit-identical to the fp32-output kernel's result cast to bf16, both engines, ragged widths, the overflow fallback too."""
    import os

    from paif_amd import ops

    B, H, W = shape
    xn = ops.to_nhwc(t(S.make_smooth_feature(79, B, 32, H, W)).to(_dev()))
    guide = ops.channel_residue(xn)
    old = os.environ.get("PAIF_GF_ENGINE")
    try:
        os.environ["PAIF_GF_ENGINE"] = engine
        for scale in (1.0, 3.0e4):                      # 3e4: a 9-row sum leaves the f16 range -> flag -> predicated VALU launch
            a = ops.guided_filter_pair(guide, xn * scale)
            b = ops.guided_filter_pair(guide, xn * scale, out_bf16=True)
            assert b.dtype == torch.bfloat16 and torch.equal(b, a.to(torch.bfloat16)), (shape, engine, scale)
    finally:
        if old is None:
            os.environ.pop("PAIF_GF_ENGINE", None)
        else:
            os.environ["PAIF_GF_ENGINE"] = old


@pytest.mark.parametrize("mengine", ["mfma2"])
def test_guided_filter_f16_range_fallback(mengine):
    """A 9-row vertical sum beyond the f16 range (65504) cannot be split into an f16 pair: the matrix-core kernels raise their
    flag and the predicated all-VALU launch behind it rewrites the output -- bit-identical to the VALU engine."""
    import os

    from paif_amd import ops

    B, H, W = 1, 40, 70
    x = t(S.make_smooth_feature(73, B, 32, H, W)).to(_dev())
    guide = ops.channel_residue(ops.to_nhwc(x))
    xn = ops.to_nhwc(x * 3.0e4)
    old = os.environ.get("PAIF_GF_ENGINE")
    try:
        os.environ["PAIF_GF_ENGINE"] = "valu"
        a = ops.guided_filter_pair(guide, xn).clone()
        os.environ["PAIF_GF_ENGINE"] = mengine
        b = ops.guided_filter_pair(guide, xn).clone()
    finally:
        if old is None:
            os.environ.pop("PAIF_GF_ENGINE", None)
        else:
            os.environ["PAIF_GF_ENGINE"] = old
    assert torch.isfinite(b).all() and torch.equal(a, b)


@pytest.mark.parametrize("shape", [(2, 24, 40), (1, 37, 53)])
def test_standalone_forwards_of_the_import_surface(shape):
    """`eca_layer.forward`, `spatial_attn_layer.forward` and the 1-argument `ChannelPool.forward` (operations_m.py:148-164, 340-367)
    called on their own -- inside the searched blocks they run fused; stand-alone they reuse the same kernels.  Against a plain torch
    fp32 restatement of the reference's lines, on ragged sizes."""
    import torch.nn.functional as F

    from paif_amd.operations_m import ChannelPool, eca_layer, spatial_attn_layer

    B, H, W = shape
    dev = _dev()
    x = t(S.make_feature(77, (B, 32, H, W), -2, 2)).to(dev)
    for k in (3, 5):
        eca = eca_layer(32, 32, 1, k_size=k)
        S.load_formula_weights(eca, salt=k)
        eca.to(dev)
        with torch.no_grad():
            y = eca(x)
            gate = torch.sigmoid(F.conv1d(x.mean((2, 3)).unsqueeze(1), eca.conv.weight, padding=(k - 1) // 2)).squeeze(1)
            ref = x * gate[:, :, None, None]
        assert y.shape == x.shape and maxabs(y.cpu(), ref.cpu()) <= 2e-6
        spa = spatial_attn_layer(k)
        S.load_formula_weights(spa, salt=10 + k)
        spa.to(dev)
        with torch.no_grad():
            z = spa(x)
            comp = torch.cat((x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)), 1)
            ref = x * torch.sigmoid(F.conv2d(comp, spa.spatial.conv.weight, padding=k // 2))
        assert maxabs(z.cpu(), ref.cpu()) <= 5e-6
    with torch.no_grad():
        c = ChannelPool()(x)
    assert tuple(c.shape) == (B, 2, H, W)
    assert maxabs(c[:, 0].cpu(), x.max(1)[0].cpu()) == 0.0 and maxabs(c[:, 1].cpu(), x.mean(1).cpu()) <= 1e-6
    with pytest.raises(NotImplementedError):
        eca_layer(32, 32, 1)(x.requires_grad_(True))            # forward-only helpers refuse to drop gradients silently


def test_guided_filter_round4_engine_long_runs_of_rows():
    """B = 20 at 480x640: 134,400 strip-rows over 128 workgroup pairs = 1,050 rows per run, more than one piece of the round-4 engine's
    per-iteration table (<= 1,000 rows): every run is walked as two or three pieces (strip boundary and table limit).  Against the
    all-VALU kernel, all 20 images."""
    import os

    from paif_amd import ops

    B, H, W = 20, 480, 640
    x1 = t(S.make_smooth_feature(83, 1, 32, H, W)).to(_dev())
    xn = ops.to_nhwc(x1).repeat(B, 1, 1, 1).contiguous()
    xn *= torch.linspace(0.5, 1.5, B, device=_dev()).view(B, 1, 1, 1)          # the images differ
    guide = ops.channel_residue(xn)
    old = os.environ.get("PAIF_GF_ENGINE")
    try:
        os.environ["PAIF_GF_ENGINE"] = "valu"
        a = ops.guided_filter_pair(guide, xn).clone()
        os.environ["PAIF_GF_ENGINE"] = "mfma2"
        b = ops.guided_filter_pair(guide, xn).clone()
    finally:
        if old is None:
            os.environ.pop("PAIF_GF_ENGINE", None)
        else:
            os.environ["PAIF_GF_ENGINE"] = old
    assert torch.isfinite(b).all()
    assert maxabs(a, b) <= 1.5e-5            # two fp32 summation orders on maps scaled up to 1.5 (measured 7.7e-6)


@pytest.mark.parametrize("storage,kh,dil,nsrc,nres,in_act,shape,kernel", [
    ("f32", 3, 1, 3, 1, 0, (2, 333, 517), "conv_bf16x3_ms<3, 1, 3, 0>"),           # RDB conv3 closing a block inside a chain
    ("f32", 3, 1, 3, 3, 0, (1, 64, 96), "conv_bf16x3_ms<3, 1, 3, 0>"),             # ... closing a chain (+ its residuals), small ragged grid
    ("f32", 3, 2, 1, 3, 2, (2, 333, 517), "conv_mfma_bf16x3<3, 2, false, 0>"),  # DilConv as one dense conv (ReLU input), tile-per-workgroup
    ("f32", 3, 1, 1, 1, 0, (2, 333, 517), "conv_bf16x3_res<3, 1, 1, 4, 0>"),       # resident-weights persistent form
    ("f32", 1, 1, 1, 0, 0, (2, 333, 517), "conv_bf16x3_ws<1, 1, 0>"),              # wave-specialised persistent form (storers pool)
    ("f16", 3, 1, 3, 3, 0, (2, 333, 517), "conv3x3_h16_dma<3, 3, 2, false, 1, 0>"),            # LDS-DMA kernel, the shipped genotype's visible chain
    ("f16", 3, 1, 3, 1, 0, (1, 480, 640), "conv3x3_h16_dma<3, 1, 2, false, 1, 0>"),
    ("bf16", 3, 1, 3, 3, 0, (2, 333, 517), "conv3x3_h16_dma<3, 3, 1, false, 1, 0>"),
    ("f16", 3, 2, 1, 3, 2, (2, 333, 517), "conv3x3_h16_dma<1, 3, 2, true, 2, 2>"),   # the shipped genotype's infrared chain (DilConv): LDS-DMA form, round 6
    ("bf16", 3, 2, 1, 3, 2, (1, 480, 640), "conv3x3_h16_dma<1, 3, 1, true, 2, 2>"),
    ("f16", 3, 2, 1, 1, 2, (2, 333, 517), "conv3x3_h16_dma<1, 1, 2, true, 2, 2>"),   # DilConv inside a chain
    ("f16", 3, 2, 1, 3, 2, (1, 64, 96), "conv_bf16x3_wsr<3, 2, 12>"),              # below the LDS-DMA kernel's tile count: the persistent kernel's storers pool
    ("f16", 3, 1, 2, 0, 0, (2, 333, 517), None),                                   # a DMA form without the fused pool: stand-alone pass behind it
    ("f16", 3, 1, 1, 0, 0, (1, 37, 53), "conv_mfma_bf16x3<3, 1, false, 12>"),
])
def test_channel_pool_fused_into_the_conv_epilogue(storage, kh, dil, nsrc, nres, in_act, shape, kernel):
    """paif_conv_desc.cpool: ChannelPool of a conv's OUTPUT (max_c, mean_c; core/model_fusion_auto.py:1352-1355) from
    its epilogue, in every kernel form that produces ir_feature / vis_feature, against the stand-alone pooling of the fp32 result.  fp32
    maps: bit-equal (same values, same summation tree).  16-bit maps: the pool is formed from the un-rounded fp32 values -- equal to the
    pool of the stored map up to that rounding."""
    from paif_amd import ops, _lib

    B, H, W = shape
    dev = _dev()
    g = torch.Generator().manual_seed(77 + kh + nsrc + nres)
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[storage]
    mk = lambda: ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), dt)
    xs, rs = [mk() for _ in range(nsrc)], [mk() for _ in range(nres)]
    w = (torch.randn(32, 32 * nsrc, kh, kh, generator=g) * 0.05).to(dev)
    prec = "f16" if storage == "f16" else "bf16x3"
    old = dict(ops.CONFIG)
    try:
        ops.set_storage(storage)
        wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision=prec)
        slope = torch.tensor([0.2], device=dev)
        kw = dict(dil=dil, in_act=in_act, act=ops.ACT_PRELU, prelu=slope, alpha=0.5, res=tuple(rs))
        ref = ops.conv2d(xs, wpk, kh, **kw)
        comp = torch.full((B, H, W, 4), float("nan"), device=dev)
        seen = []
        orig = ops.channel_pool1
        ops.channel_pool1 = lambda *a, **k: (seen.append(1), orig(*a, **k))[1]
        try:
            out = ops.conv2d(xs, wpk, kh, cpool=(comp, 2), **kw)
        finally:
            ops.channel_pool1 = orig
        torch.cuda.synchronize()
        assert bool(seen) == (kernel is None), "fused where the kernel can, the stand-alone pass otherwise"
        assert torch.equal(out, ref)
        assert bool(torch.isnan(comp[..., :2]).all()), "only the addressed half of the plane is written"
        r32 = ref.float()
        want = torch.stack([r32.max(-1).values, r32.mean(-1)], -1)
        got = comp[..., 2:]
        if storage == "f32":
            pooled = torch.empty((B, H, W, 4), device=dev)
            ops.channel_pool1(ref, pooled, 2)
            assert torch.equal(got, pooled[..., 2:])
        if kernel is None:
            assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
        else:   # from the un-rounded values: within the output format's rounding of the pooled stored map
            eps = {"f32": 1e-6, "f16": 2.0 ** -11, "bf16": 2.0 ** -8}[storage]
            assert float((got - want).abs().max()) <= 1.01 * eps * float(r32.abs().max()) + 1e-6
    finally:
        ops.CONFIG.update(old)


@pytest.mark.parametrize("storage", ["f32", "f16", "bf16"])
def test_fusion_forward_with_and_without_the_fused_channel_pool(storage):
    """The inference forward with ChannelPool(ir_feature, vis_feature) fused into the two chains' last convs vs the stand-alone
    channel_pool2 pass (ops.CONFIG['cpool_fused'] = False): fp32 storage bit-equal, 16-bit storage equal up to the pool's use of un-rounded
    values (the blend weight sigmoid(conv5x5(pool)) moves by ~1e-4)."""
    from paif_amd import ops

    net = _fusion_net()
    ir, vis, _ = S.make_batch(2, 64, 96)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    old = dict(ops.CONFIG)
    try:
        ops.set_storage(storage)
        calls = []
        orig = ops.channel_pool2
        ops.channel_pool2 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            with torch.no_grad():
                fused = net(irt, ycc)
                assert not calls, "channel_pool2 still launched in the inference forward"
                ops.CONFIG["cpool_fused"] = False
                plain = net(irt, ycc)
                assert calls
        finally:
            ops.channel_pool2 = orig
        if storage == "f32":
            assert torch.equal(fused, plain)
        else:
            d = float((fused - plain).abs().max())
            assert d <= (2e-3 if storage == "f16" else 1.5e-2), d      # = the storage mode's own distance from the fp32 forward
    finally:
        ops.CONFIG.update(old)
