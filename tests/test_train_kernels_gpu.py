"""GPU unit parity of the training-step kernels (csrc/train_kernels.hip, through the C ABI) against torch's CPU ops and
autograd -- the plain references of the same operators the reference model is made of (nn.BatchNorm2d in train mode,
nn.PReLU, depthwise nn.Conv2d, torch.optim.AdamW ...).  Shapes are odd on purpose (ragged row blocks)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from paif_amd import ops, synthetic as S
from tests.helpers import t, maxabs

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _nhwc(x):
    return ops.to_nhwc(x.to(_dev()))


def _rel(a, b):
    return maxabs(a, b) / max(1e-30, float(b.abs().max()))


@pytest.mark.parametrize("C,B,H,W", [(32, 3, 37, 53), (256, 2, 30, 41), (32, 1, 7, 9)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_train_mode_batchnorm_forward_backward(C, B, H, W, act):
    """nn.BatchNorm2d(train) (+ PReLU / ReLU) forward incl. the running-statistics update, and its backward (dx, dgamma,
    dbeta, dslope) -- operations_m.py:458-460,503; core/segformer_head.py:50-55."""
    g = torch.Generator().manual_seed(C + B + act)
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    slope = torch.tensor([0.2], requires_grad=True)
    bn.train()
    z = bn(x)
    y = F.prelu(z, slope) if act == 1 else (F.relu(z) if act == 2 else z)
    dy = torch.randn(B, C, H, W, generator=g)
    (y * dy).sum().backward()
    dev = _dev()
    xn = _nhwc(x.detach())
    rm, rv = rm0.to(dev), rv0.to(dev)
    gam, bet, sl = bn.weight.detach().to(dev), bn.bias.detach().to(dev), slope.detach().to(dev)
    stats = ops.bn_stats(xn, gam, bet, bn.eps, bn.momentum, rm, rv)
    out = ops.affine_act_res(xn, stats[2], stats[3], act, sl if act == 1 else None)
    assert _rel(out.permute(0, 3, 1, 2).cpu(), y.detach()) <= 3e-6
    assert maxabs(rm.cpu(), bn.running_mean) <= 1e-6 and _rel(rv.cpu(), bn.running_var) <= 1e-6
    dg, db, ds = (torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(1, device=dev))
    dx = ops.bn_act_bwd(_nhwc(dy), xn, stats, act, sl if act == 1 else None, dg, db, ds)
    assert _rel(dx.permute(0, 3, 1, 2).cpu(), x.grad) <= 2e-5
    assert _rel(dg.cpu(), bn.weight.grad) <= 2e-5 and _rel(db.cpu(), bn.bias.grad) <= 2e-5
    if act == 1:
        assert _rel(ds.cpu(), slope.grad) <= 2e-5
    # gradients ACCUMULATE: a second call doubles them
    ops.bn_act_bwd(_nhwc(dy), xn, stats, act, sl if act == 1 else None, dg, db, ds)
    assert _rel(dg.cpu(), 2 * bn.weight.grad) <= 2e-5


def test_prelu_backward_and_tail_dz():
    g = torch.Generator().manual_seed(5)
    r = torch.randn(2, 32, 19, 23, generator=g, requires_grad=True)
    slope = torch.tensor([0.15], requires_grad=True)
    tt = torch.randn(2, 32, 19, 23, generator=g)
    add = torch.randn(2, 32, 19, 23, generator=g)
    (F.prelu(r, slope) * tt).sum().backward()
    dev = _dev()
    ds = torch.zeros(1, device=dev)
    dx = ops.prelu_bwd(_nhwc(tt), _nhwc(r.detach()), slope.detach().to(dev), ds, add=_nhwc(add), want_dx=True, factor=0.5)
    assert _rel(dx.permute(0, 3, 1, 2).cpu(), r.grad + add) <= 1e-6
    assert _rel(ds.cpu(), 0.5 * slope.grad) <= 1e-5
    # tail: fused = tanh(PReLU(z))
    z = torch.randn(2, 1, 33, 41, generator=g, requires_grad=True)
    slope2 = torch.tensor([0.3], requires_grad=True)
    fused = torch.tanh(F.prelu(z, slope2))
    dfu = torch.randn(2, 1, 33, 41, generator=g)
    (fused * dfu).sum().backward()
    ds2 = torch.zeros(1, device=dev)
    dz = ops.tail_dz(dfu.to(dev), fused.detach().to(dev), z.detach().to(dev), slope2.detach().to(dev), ds2)
    assert _rel(dz.cpu(), z.grad) <= 2e-6 and _rel(ds2.cpu(), slope2.grad) <= 1e-5


@pytest.mark.parametrize("C,k,dil,relu,bias", [(32, 3, 2, True, False), (32, 3, 1, False, False), (128, 3, 1, False, True),
                                               (640, 3, 1, False, True), (32, 5, 1, True, False), (2048, 3, 1, False, True)])
def test_depthwise_conv_weight_gradient(C, k, dil, relu, bias):
    """DilConv's depthwise (ReLU in front, operations_m.py:496-498) and MiT's Mlp.dwconv (+bias, core/mix_transformer.py:376-387)."""
    B, H, W = (2, 17, 21) if C <= 640 else (1, 9, 11)
    g = torch.Generator().manual_seed(C + k)
    x = torch.randn(B, C, H, W, generator=g)
    w = (torch.randn(C, 1, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True) if bias else None
    dy = torch.randn(B, C, H, W, generator=g)
    y = F.conv2d(F.relu(x) if relu else x, w, b, 1, dil * (k - 1) // 2, dil, C)
    (y * dy).sum().backward()
    dev = _dev()
    dw = torch.zeros(C, 1, k, k, device=dev)
    db = torch.zeros(C, device=dev) if bias else None
    ops.dwconv_wgrad(_nhwc(x), _nhwc(dy), k, dil, relu, dw, db)
    assert _rel(dw.cpu(), w.grad) <= 2e-5
    if bias:
        assert _rel(db.cpu(), b.grad) <= 2e-5


def test_stem_and_small_conv_weight_gradients():
    g = torch.Generator().manual_seed(9)
    dev = _dev()
    # stem: Conv2d(1,32,3,pad 1) + PReLU on channel 0 of a [B,3,H,W] tensor (the Y plane of YCrCb) -- batch-strided view
    img3 = torch.rand(2, 3, 31, 45, generator=g)
    w = (torch.randn(32, 1, 3, 3, generator=g) * 0.4).requires_grad_(True)
    slope = torch.tensor([0.2], requires_grad=True)
    dfeat = torch.randn(2, 32, 31, 45, generator=g)
    (F.prelu(F.conv2d(img3[:, 0:1], w, None, 1, 1), slope) * dfeat).sum().backward()
    dw, ds = torch.zeros(32, 1, 3, 3, device=dev), torch.zeros(1, device=dev)
    ops.stem_wgrad(img3.to(dev)[:, 0:1], _nhwc(dfeat), w.detach().to(dev), slope.detach().to(dev), dw, ds)
    assert _rel(dw.cpu(), w.grad) <= 2e-5 and _rel(ds.cpu(), slope.grad) <= 2e-5
    # Cm -> 1 convs: stem_out.1 (16 -> 1, k 3) and spatial_attn_layer_M (4 -> 1, k 5)
    for Cm, k in ((16, 3), (4, 5)):
        m = torch.randn(2, Cm, 29, 37, generator=g)
        wc = (torch.randn(1, Cm, k, k, generator=g) * 0.2).requires_grad_(True)
        s = torch.randn(2, 1, 29, 37, generator=g)
        (F.conv2d(m, wc, None, 1, k // 2) * s).sum().backward()
        dwc = torch.zeros(1, Cm, k, k, device=dev)
        ops.corr1_wgrad(s.to(dev), _nhwc(m), k, dwc)
        assert _rel(dwc.cpu(), wc.grad) <= 2e-5, (Cm, k)
    # column sums (conv bias gradient)
    x = torch.randn(3, 17, 19, 32, generator=g)
    out = torch.ones(32, device=dev)
    ops.colsum(x.to(dev), out)
    assert _rel(out.cpu(), 1 + x.reshape(-1, 32).double().sum(0).float()) <= 1e-6


def test_weight_gradient_layout_helpers():
    g = torch.Generator().manual_seed(11)
    dev = _dev()
    # Cell_Decom 1x1: the folded gradient G over [x, LF1, LF2] -> the 128-channel weight [LF1, LF2, x-LF1, x-LF2]
    x, l1, l2 = (torch.randn(2, 32, 13, 15, generator=g) for _ in range(3))
    w = (torch.randn(32, 128, 1, 1, generator=g) * 0.1).requires_grad_(True)
    dy = torch.randn(2, 32, 13, 15, generator=g)
    (F.conv2d(torch.cat([l1, l2, x - l1, x - l2], 1), w) * dy).sum().backward()
    G = ops.conv2d_wgrad([_nhwc(x), _nhwc(l1), _nhwc(l2)], _nhwc(dy), 1)
    dw = torch.zeros(32, 128, 1, 1, device=dev)
    ops.unfold_decomp1x1_wgrad(G, dw)
    assert _rel(dw.cpu(), w.grad) <= 2e-5
    # conv-as-GEMM: patch-embed 7x7 stride 4 on 3 channels, SR conv 2x2 stride 2
    for Cin, Cout, k, st, pad in ((3, 32, 7, 4, 3), (64, 64, 2, 2, 0), (32, 64, 3, 2, 1)):
        xi = torch.randn(2, Cin, 24, 32, generator=g)
        wc = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_(True)
        bc = torch.randn(Cout, generator=g).requires_grad_(True)
        y = F.conv2d(xi, wc, bc, st, pad)
        dyc = torch.randn(y.shape, generator=g)
        (y * dyc).sum().backward()
        wp = ops.pack_conv_gemm_weight(wc.detach().to(dev))
        col = ops.im2col(_nhwc(xi), k, st, pad, wp.shape[1])
        dwp, dbp = ops.gemm_wgrad(_nhwc(dyc).reshape(-1, Cout), col.reshape(-1, wp.shape[1]))
        dwc = torch.zeros(Cout, Cin, k, k, device=dev)
        ops.unpack_conv_gemm_wgrad(dwp, dwc)
        assert _rel(dwc.cpu(), wc.grad) <= 3e-5 and _rel(dbp.cpu(), bc.grad) <= 3e-5, (Cin, Cout, k)
    # dense conv, 32 -> 16 (stem_out.0): zero-padded dout, first 16 rows, accumulated twice
    xi = torch.randn(2, 32, 14, 18, generator=g)
    wc = (torch.randn(16, 32, 3, 3, generator=g) * 0.1).requires_grad_(True)
    d16 = torch.randn(2, 16, 14, 18, generator=g)
    (F.conv2d(xi, wc, None, 1, 1) * d16).sum().backward()
    acc = torch.zeros(16, 32, 3, 3, device=dev)
    for _ in range(2):
        ops.conv2d_wgrad([_nhwc(xi)], ops.pad_channels(_nhwc(d16), 32), 3, out=acc, cout=16)
    assert _rel(acc.cpu(), 2 * wc.grad) <= 2e-5


def test_keep_masks_replay_on_the_host_and_scaling():
    dev = _dev()
    rng = ops.DropRNG(1234)
    rng.reseed(1234, rank=3, step=17)
    seed = rng.seed
    m1 = rng.keep_mask(64, 0.1, dev)
    m2 = rng.keep_mask(1000, 0.3, dev)
    for m, off, p in ((m1, 0, 0.1), (m2, 64, 0.3)):
        u = S.hash_uniform(seed, m.numel(), offset=off)
        ref = np.where(u >= np.float64(np.float32(p)), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)
        assert np.array_equal(m.cpu().numpy(), ref)
    assert 0.6 < float((m2 > 0).float().mean()) < 0.8
    x = torch.randn(3, 50, 64)
    res = torch.randn(3, 50, 64)
    s1, s2 = torch.rand(3), torch.rand(3, 64)
    a = ops.rowscale_add(x.to(dev), s1.to(dev), res.to(dev))
    b = ops.rowscale_add(x.to(dev), s2.to(dev), None, per_channel=True)
    assert maxabs(a.cpu(), x * s1.view(3, 1, 1) + res) <= 1e-6 and maxabs(b.cpu(), x * s2.view(3, 1, 64)) <= 1e-6


def test_upsample_ce_autograd_node_matches_torch():
    g = torch.Generator().manual_seed(21)
    seg = torch.randn(2, 9, 24, 32, generator=g, requires_grad=True)
    lab = torch.randint(0, 9, (2, 96, 128), generator=g)
    lab[0, :5] = 255
    up = F.interpolate(seg, size=(96, 128), mode="bilinear", align_corners=False)
    ref = F.cross_entropy(up, lab, ignore_index=255)
    (ref * 3.0).backward()
    dev = _dev()
    sd = seg.detach().to(dev).requires_grad_(True)
    mine = ops.upsample_ce(sd, lab.to(dev))
    (mine * 3.0).backward()
    assert abs(float(mine) - float(ref)) <= 1e-5 * abs(float(ref))
    assert _rel(sd.grad.cpu(), seg.grad) <= 2e-5


def test_colour_and_decomposition_public_forms(golden):
    """F2 stand-alone YCrCb2RGB (core/model_fusion_auto.py:94-111) and Cell_Decom.decomposition (:522-535)."""
    from paif_amd.core.model_fusion_auto import YCrCb2RGB, Network_Fusion_Searched
    from paif_amd.genotypes import FUSION_AT

    g = golden("gd_colour_glue")
    dev = _dev()
    assert maxabs(YCrCb2RGB(t(g["ycc"]).to(dev)).cpu(), g["rgb"]) <= 1e-6
    gj = golden("gj_showfeatures_40x56")
    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    S.load_formula_weights(net)
    net.to(dev)
    ir, vis, _ = S.make_batch(1, 40, 56)
    with torch.no_grad():
        fir, _ = ops.stem(t(ir).to(dev), net.stem_1[0].weight, net.stem_1[1].weight)
        lf, hf = net.decompation.decomposition(ops.to_nchw_view(fir))
    assert tuple(lf.shape) == (1, 64, 40, 56)
    assert maxabs(lf.cpu(), gj["lf_ir"]) <= 1e-4 and maxabs(hf.cpu(), gj["hf_ir"]) <= 1e-4


def test_byte_upload_conversions_are_bit_exact():
    g = torch.Generator().manual_seed(31)
    v8 = torch.randint(0, 256, (2, 20, 30, 3), generator=g, dtype=torch.uint8)
    l8 = torch.randint(0, 9, (2, 20, 30), generator=g, dtype=torch.uint8)
    dev = _dev()
    f = ops.u8_to_planes(v8.to(dev)).cpu().numpy()
    ref = v8.numpy().transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(f, ref)
    assert np.array_equal(ops.u8_to_i64(l8.to(dev)).cpu().numpy(), l8.numpy().astype(np.int64))
