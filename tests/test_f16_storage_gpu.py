"""GPU tests of the fp16 ACTIVATION-STORAGE mode (round 5; `ops.set_storage("f16")`): the 16-bit configuration built to meet SURVEY
8(d)'s bf16-clause (report max / mean |fused - reference|, argmax agreement >= 99.9 %, mIoU within 0.1 pt) that the bf16 one misses.

IEEE fp16 maps (11 significant bits) behind the guided-filter block, fp16 weights, one `v_mfma_f32_32x32x16_f16` per product, fp32
accumulate; the guided filter writes HF = x - LF and the 1x1 behind it folds over [x, HF1, HF2]; the forward's last 32-channel map
stays fp32 (tools/storage_sensitivity.py, tools/f16_ablation.py: why).  Unit level: every kernel form, on fp16-representable inputs and weights
(products then exact), must reproduce the fp32-storage kernel's result up to the rounding of its own output (2^-11 relative)."""
import json
import os

import numpy as np
import pytest
import torch

from paif_amd import ops, synthetic as S
from tests import helpers as Hh
from tests.helpers import t, maxabs

pytestmark = pytest.mark.gpu
H_EPS = 2.0 ** -11          # half an ulp of an fp16 value, relative
F16 = torch.float16


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_arithmetic():
    old = dict(ops.CONFIG)
    ops.set_conv_precision("bf16x3")
    ops.set_storage("f32")
    yield
    ops.CONFIG.update(old)
    ops._ACT_BF16[0] = False


def _rounded(x):
    """fp32 map whose values are exactly representable in fp16, and its fp16 twin."""
    xh = ops.cast_storage(x, F16)
    return ops.cast_storage(xh, torch.float32), xh


def _close(out, ref, frac=0.0, abs_floor=1e-5):
    """|out - ref| within half an fp16 ulp of ref (+ the fp32 accumulation noise where terms cancel); frac > 0: that fraction of the
    elements may land on the other fp16 neighbour (another accumulation order than the reference kernel's)."""
    err = (out.float() - ref).abs()
    tol = H_EPS * ref.abs() * 1.01 + abs_floor
    if frac == 0.0:
        assert bool((err <= tol).all()), (float(err.max()), float((err / ref.abs().clamp_min(1e-6)).max()))
    else:
        bad = err > tol
        assert float(bad.float().mean()) < frac and bool((err <= 2 * tol).all()), (int(bad.sum()), float(err.max()))


def test_cast_round_trip_is_round_to_nearest_even():
    x = t(S.make_feature(5, (3, 7, 9, 32), -4, 4)).to(_dev())
    x[0, 0, 0, :4] = torch.tensor([1e-6, -3e-8, 70000.0, 6.1e-5], device=x.device)      # subnormal, underflow, overflow -> inf, smallest normal
    xh = ops.cast_storage(x, F16)
    assert xh.dtype == F16 and torch.equal(xh, x.to(F16))          # torch's cast is RNE
    assert torch.equal(ops.cast_storage(xh, torch.float32), xh.float())
    with pytest.raises(TypeError):
        ops.cast_storage(xh, torch.bfloat16)


def _conv_case(kh, dil, nsrc, nres, cout, B, H, W, seed, wscale=0.05):
    g = torch.Generator().manual_seed(seed)
    dev = _dev()
    xs32, xsh, rs32, rsh = [], [], [], []
    for _ in range(nsrc):
        a, b = _rounded(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)))
        xs32.append(a); xsh.append(b)
    for _ in range(nres):
        a, b = _rounded(ops.to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev)))
        rs32.append(a); rsh.append(b)
    w = (torch.randn(cout, 32 * nsrc, kh, kh, generator=g) * wscale).to(dev).to(F16).float()      # fp16-representable: exact products
    scale, shift = (torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
    return xs32, xsh, rs32, rsh, w, scale, shift, torch.tensor([0.2], device=dev)


@pytest.mark.parametrize("kh,dil,nsrc,nres,act,cout,kernel", [
    (3, 1, 1, 0, 1, 32, "conv3x3_h16_dma<1, 0, 2, false, 1, 0>"),
    (3, 1, 1, 1, 0, 32, "conv3x3_h16_dma<1, 1, 2, false, 1, 0>"),
    (3, 1, 2, 0, 1, 32, "conv3x3_h16_dma<2, 0, 2, false, 1, 0>"),
    (3, 1, 3, 1, 1, 32, "conv3x3_h16_dma<3, 1, 2, false, 1, 0>"),
    (3, 1, 3, 3, 2, 32, "conv3x3_h16_dma<3, 3, 2, false, 1, 0>"),
    (3, 1, 1, 0, 0, 16, "conv3x3_h16_dma<1, 0, 2, false, 1, 0>"),          # 32 -> 16 (stem_out.0 of the two-kernel tail)
    (7, 1, 1, 0, 1, 32, "conv7x7_h16_dma<2>"),
    (1, 1, 1, 0, 0, 32, "conv_bf16x3_ws<1, 1, 12>"),           # 1x1 stream (wave-specialised)
    (1, 1, 1, 2, 1, 32, "conv_mfma_bf16x3<1, 1, false, 12>"),   # 1x1 with residual maps (tile-per-workgroup kernel)
    (3, 2, 1, 0, 0, 32, "conv_bf16x3_ws<3, 2, 12>"),           # dilation 2 (wave-specialised)
    (3, 2, 1, 2, 1, 32, "conv_bf16x3_ws<3, 2, 12>"),
])
@pytest.mark.parametrize("shape", [(2, 333, 517), (1, 480, 640)])
def test_dense_conv_f16_storage_matches_fp32_storage(kh, dil, nsrc, nres, act, cout, kernel, shape):
    B, H, W = shape
    xs32, xsh, rs32, rsh, w, scale, shift, slope = _conv_case(kh, dil, nsrc, nres, cout, B, H, W, kh * 100 + dil * 10 + nsrc + 7 * nres,
                                                              0.02 if kh == 7 else 0.05)
    kw = dict(scale=scale, shift=shift, act=act, prelu=slope if act == 1 else None, alpha=0.5, cout=cout)
    ref = ops.conv2d(xs32, ops.pack_conv_weight(w, nsrc, 32, kh, precision="bf16x3"), kh, dil=dil, res=tuple(rs32), **kw)
    wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision="f16")
    from paif_amd import _lib
    d = _lib.ConvDesc()
    d.storage, d.precision, d.nsrc, d.cin, d.kh, d.dil, d.cout, d.alpha = 3, 4, nsrc, 32, kh, dil, cout, 0.5
    for i, r in enumerate(rsh):
        d.res[i] = ops._pa(r)
    assert ops.conv2d_kernel_name(d, B, H, W) == kernel
    out = ops.conv2d(xsh, wpk, kh, dil=dil, res=tuple(rsh), **kw)
    torch.cuda.synchronize()
    assert out.dtype == F16 and ref.dtype == torch.float32
    _close(out, ref, frac=1e-4)


@pytest.mark.parametrize("fmt", ["f16", "bf16"])
@pytest.mark.parametrize("nres", [1, 3])
@pytest.mark.parametrize("shape", [(2, 333, 517), (1, 480, 640), (4, 96, 100)])
def test_dilated_conv_behind_a_relu_on_the_lds_dma_kernel(shape, nres, fmt):
    """Round 6: the composed DilConv -- input ReLU, dense 3x3 dilation 2, affine, 1 or 3 residual maps
    (operations_m.py:494-506 as one conv) -- on the LDS-DMA kernel conv3x3_h16_dma<1, NRES, F, CP, 2, 2>: 12 x 36 halo tiles, the wave's
    two output rows two apart (four A fragments feed six MFMAs at dilation 2 as well), ReLU as v_pk_max_i16 on the fragments.  Against the
    fp32-storage conv on the same (16-bit-representable) values; ragged tiles; both formats."""
    B, H, W = shape
    dt = F16 if fmt == "f16" else torch.bfloat16
    g = torch.Generator().manual_seed(1000 + nres + W)
    dev = _dev()
    x32 = ops.cast_storage(ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), dt), torch.float32)
    r32 = [ops.cast_storage(ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), dt), torch.float32) for _ in range(nres)]
    xh, rh = ops.cast_storage(x32, dt), [ops.cast_storage(r, dt) for r in r32]
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).to(dev).to(dt).float()
    scale, shift = (torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
    kw = dict(dil=2, in_act=ops.ACT_RELU, scale=scale, shift=shift)
    ref = ops.conv2d([x32], ops.pack_conv_weight(w, 1, 32, 3, precision="bf16x3"), 3, res=tuple(r32), **kw)
    from paif_amd import _lib
    old = dict(ops.CONFIG)
    try:
        ops.set_storage(fmt)
        wpk = ops.pack_conv_weight(w, 1, 32, 3, precision="f16" if fmt == "f16" else "bf16x3")
        d = _lib.ConvDesc()
        d.storage, d.precision, d.nsrc, d.cin, d.kh, d.dil, d.cout, d.alpha, d.in_act = (3 if fmt == "f16" else 1), (4 if fmt == "f16" else 2), 1, 32, 3, 2, 32, 1.0, 2
        for i, r in enumerate(rh):
            d.res[i] = ops._pa(r)
        big = B * ((H + 7) // 8) * ((W + 31) // 32) >= 1024
        name = ops.conv2d_kernel_name(d, B, H, W)
        assert name == ("conv3x3_h16_dma<1, %d, %d, false, 2, 2>" % (nres, 2 if fmt == "f16" else 1) if big else name), name
        assert big or "h16_dma" not in name
        out = ops.conv2d([xh], wpk, 3, res=tuple(rh), **kw)
        torch.cuda.synchronize()
    finally:
        ops.CONFIG.update(old)
    assert out.dtype == dt
    eps = H_EPS if fmt == "f16" else 2.0 ** -8
    err = (out.float() - ref).abs()
    tol = eps * ref.abs() * 1.01 + 1e-5
    bad = err > tol
    assert float(bad.float().mean()) < 1e-4 and bool((err <= 2 * tol).all()), (int(bad.sum()), float(err.max()))


@pytest.mark.parametrize("kh,dil,nsrc,nres,act,cout", [(3, 1, 1, 0, 1, 32), (3, 1, 2, 1, 1, 32), (3, 1, 3, 2, 0, 32), (1, 1, 3, 0, 0, 32),
                                                      (3, 2, 1, 1, 1, 32), (7, 1, 1, 0, 0, 32), (3, 1, 1, 0, 0, 16), (5, 1, 1, 0, 0, 32)])
def test_dense_conv_f16_storage_small_and_ragged_shapes(kh, dil, nsrc, nres, act, cout):
    """Shapes below the persistent kernels' tile counts (the 64x96 parity cases, ragged edges): the tile-per-workgroup kernel with fp16
    operands, every source count.  5x5 is not built for 16-bit maps (no searched cell of the shipped genotype uses it): it must say so."""
    B, H, W = 2, 37, 53
    xs32, xsh, rs32, rsh, w, scale, shift, slope = _conv_case(kh, dil, nsrc, nres, cout, B, H, W, 900 + kh * 10 + nsrc)
    kw = dict(scale=scale, shift=shift, act=act, prelu=slope if act == 1 else None, alpha=0.5, cout=cout)
    wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision="f16")
    if kh == 5:
        with pytest.raises(RuntimeError, match="not built"):
            ops.conv2d(xsh, wpk, kh, dil=dil, res=tuple(rsh), **kw)
        return
    ref = ops.conv2d(xs32, ops.pack_conv_weight(w, nsrc, 32, kh, precision="bf16x3"), kh, dil=dil, res=tuple(rs32), **kw)
    out = ops.conv2d(xsh, wpk, kh, dil=dil, res=tuple(rsh), **kw)
    assert out.dtype == F16
    _close(out, ref, frac=1e-4)


@pytest.mark.parametrize("shape", [(2, 333, 517), (2, 37, 53)])
def test_folded_decomposition_1x1_with_two_piece_weights(shape):
    """The 1x1 behind the guided filter in the fp16 forward: three fp16 sources, weights as fp16 hi + lo (precision "f16x2", two MFMAs per
    product): ARBITRARY fp32 weights are reproduced to 2^-22 -- the result equals the fp32-storage kernel's up to the output rounding."""
    B, H, W = shape
    xs32, xsh, _, _, _, _, shift, _ = _conv_case(1, 1, 3, 0, 32, B, H, W, 4321)
    g = torch.Generator().manual_seed(99)
    w = (torch.randn(32, 96, 1, 1, generator=g) * 0.1).to(_dev())              # NOT fp16-representable
    ref = ops.conv2d(xs32, ops.pack_conv_weight(w, 3, 32, 1, precision="bf16x3"), 1, shift=shift)
    out = ops.conv2d(xsh, ops.pack_conv_weight(w, 3, 32, 1, precision="f16x2"), 1, shift=shift)
    assert out.dtype == F16
    _close(out, ref, frac=1e-4, abs_floor=2e-5)           # (the split-bf16 reference itself carries 2^-17 per product)
    one = ops.conv2d(xsh, ops.pack_conv_weight(w, 3, 32, 1, precision="f16"), 1, shift=shift)      # hi pieces only: visibly coarser
    assert float((one.float() - ref).abs().mean()) > 1.3 * float((out.float() - ref).abs().mean())     # (both include the output rounding)
    assert float((one != out).float().mean()) > 0.2


def test_decomposition_fold_over_high_frequency_maps():
    """paif_pack_decomp1x1_hf_weight_f16x2: W . [LF1, LF2, x - LF1, x - LF2] as a 1x1 over [x, HF1, HF2] with HF_i = x - LF_i
    (core/model_fusion_auto.py:512-513, :531-535), against the reference-order evaluation in float64."""
    dev = _dev()
    B, H, W = 1, 24, 40
    g = torch.Generator().manual_seed(5)
    x32, xh = _rounded(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)))
    hf = [_rounded((torch.randn(B, H, W, 32, generator=g) * 0.05).to(dev)) for _ in range(2)]
    w = (torch.randn(32, 128, 1, 1, generator=g) * 0.1).to(dev)
    bias = torch.randn(32, generator=g).to(dev)
    ops.CONFIG["f16_decomp_split"] = True      # the hi + lo pack: arbitrary fp32 weights to 2^-22 (the default reads the hi pieces only)
    out = ops.conv2d([xh, hf[0][1], hf[1][1]], ops.pack_decomp1x1_hf_weight(w), 1, shift=bias)
    ops.CONFIG["f16_decomp_split"] = False
    one = ops.conv2d([xh, hf[0][1], hf[1][1]], ops.pack_decomp1x1_hf_weight(w), 1, shift=bias)
    assert float((one.float() - out.float()).abs().max()) <= 4e-3
    x64 = x32.double()
    lf = [x64 - h[0].double() for h in hf]
    cat = torch.cat([lf[0], lf[1], x64 - lf[0], x64 - lf[1]], dim=-1)                       # [LF1, LF2, HF1, HF2]
    ref = (cat @ w.double().reshape(32, 128).t() + bias.double()).float()
    _close(out, ref, abs_floor=2e-6)


@pytest.mark.parametrize("kh", [1, 3])
@pytest.mark.parametrize("pool", [False, True])
def test_dense_conv_f16_storage_input_prelu(kh, pool):
    """Input PReLU on an fp16-stored source (ECABasicBlock's second conv, operations_m.py:383-385; with the ECA pool partials): the
    PReLU result is rounded to fp16 (nearest-even) like every other operand -> equal, to the output rounding, to the fp32 kernel fed
    that rounded map."""
    if pool and kh == 1:
        pytest.skip("the pooled form is the 3x3 of the ECA block")
    B, H, W = 2, 141, 203
    xs32, xsh, _, _, w, _, _, slope = _conv_case(kh, 1, 1, 0, 32, B, H, W, 17 + kh)
    xin = torch.where(xs32[0] >= 0, xs32[0], xs32[0] * slope).to(F16).float()
    r = ops.conv2d([xin], ops.pack_conv_weight(w, 1, 32, kh, precision="bf16x3"), kh, pool=pool)
    o = ops.conv2d(xsh, ops.pack_conv_weight(w, 1, 32, kh, precision="f16"), kh, in_act=ops.ACT_PRELU, in_prelu=slope, pool=pool)
    if pool:
        (ref, pref), (out, pout) = r, o
        assert float((pout - pref).abs().max()) <= 2e-3 * float(pref.abs().max()) + 0.05     # tile sums of the ROUNDED outputs vs of the fp32 ones
    else:
        ref, out = r, o
    assert out.dtype == F16
    _close(out, ref, frac=1e-4)


@pytest.mark.parametrize("shape", [(2, 333, 517), (2, 37, 53)])
@pytest.mark.parametrize("nres", [0, 2])
def test_last_conv_writes_fp32_from_fp16_maps(shape, nres):
    """PAIF_ST_F16_F32: the dilation-2 3x3 that produces the forward's last 32-channel map reads fp16 sources and residual maps and
    writes fp32 -- equal to the fp32-storage kernel up to fp32 summation noise (NO output rounding)."""
    B, H, W = shape
    xs32, xsh, rs32, rsh, w, scale, shift, slope = _conv_case(3, 2, 1, nres, 32, B, H, W, 555 + nres)
    kw = dict(scale=scale, shift=shift, act=1, prelu=slope)
    ref = ops.conv2d(xs32, ops.pack_conv_weight(w, 1, 32, 3, precision="bf16x3"), 3, dil=2, res=tuple(rs32), **kw)
    out = ops.conv2d(xsh, ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3, dil=2, res=tuple(rsh), out_f32=True, **kw)
    assert out.dtype == torch.float32
    assert maxabs(out, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # a request, not a demand: where no such kernel exists the map comes back in the sources' format
    o2 = ops.conv2d(xsh, ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3, dil=2, res=tuple(rsh), in_act=ops.ACT_RELU, out_f32=True, **kw)
    assert o2.dtype == F16


def test_storage_and_pack_precision_must_agree():
    xs32, xsh, _, _, w, _, _, _ = _conv_case(3, 1, 1, 0, 32, 1, 16, 32, 1)
    with pytest.raises(RuntimeError, match="fp16 maps need an fp16 weight pack"):
        ops.conv2d(xsh, ops.pack_conv_weight(w, 1, 32, 3, precision="bf16x3"), 3)
    with pytest.raises(RuntimeError, match="fp16 maps need an fp16 weight pack"):
        ops.conv2d(xs32, ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3)
    with pytest.raises(NotImplementedError):
        ops.conv2d(xsh, ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3, want_aux=True)


@pytest.mark.parametrize("dt", [F16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 480, 640), (3, 4, 64), (1, 5, 130), (2, 64, 96)])
def test_spatial_blend_tiled_kernel_is_bit_equal(shape, dt, monkeypatch):
    """Round 6: spa_blend on 16-bit maps as a 4 x 64 tile kernel (pooled patch through LDS, 16-byte map pieces) -- the same partial sums and
    the same summation tree as the pixel-per-8-lanes kernel: bit-identical maps, including ragged tiles."""
    B, H, W = shape
    dev = _dev()
    g = torch.Generator().manual_seed(17 + W)
    a = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)).to(dt)
    b = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)).to(dt)
    comp = ops.channel_pool2(a.float(), b.float())
    w5 = (torch.randn(1, 4, 5, 5, generator=g) * 0.2).to(dev)
    monkeypatch.setenv("PAIF_SPA_TILED", "0")
    ref = ops.spa_blend(comp, w5, a, b)
    monkeypatch.setenv("PAIF_SPA_TILED", "1")
    out = ops.spa_blend(comp, w5, a, b)
    torch.cuda.synchronize()
    assert out.dtype == dt and torch.equal(out, ref)


def test_elementwise_kernels_f16_storage():
    dev = _dev()
    B, H, W = 2, 37, 53
    g = torch.Generator().manual_seed(3)
    a32, ah = _rounded(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)))
    b32, bh = _rounded(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)))

    def close(xh, x32):
        err = (xh.float() - x32).abs()
        return xh.dtype == F16 and bool((err <= H_EPS * x32.abs() * 1.01 + 1e-7).all())

    wd = (torch.randn(32, 1, 3, 3, generator=g) * 0.3).to(dev)
    for dil in (1, 2):
        assert close(ops.dwconv(ah, wd, 3, dil, in_relu=True), ops.dwconv(a32, wd, 3, dil, in_relu=True))
    w5d = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).to(dev)
    assert close(ops.dwconv(ah, w5d, 5, 1, in_relu=False), ops.dwconv(a32, w5d, 5, 1, in_relu=False))
    assert torch.equal(ops.channel_pool2(ah, bh), ops.channel_pool2(a32, b32))                 # fp32 output from identical values
    w5 = (torch.randn(1, 4, 5, 5, generator=g) * 0.2).to(dev)
    comp = ops.channel_pool2(a32, b32)
    assert close(ops.spa_blend(comp, w5, ah, bh), ops.spa_blend(comp, w5, a32, b32))
    assert close(ops.add(ah, bh), ops.add(a32, b32))
    assert close(ops.add(ah, b32), ops.add(a32, b32))                                          # mixed: the fp32 map is rounded first (it is representable)
    part = torch.rand(ops.lib().paif_conv2d_blocks(B, H, W), 32, device=dev)
    w1d = torch.randn(3, device=dev)
    slope = torch.tensor([0.25], device=dev)
    assert close(ops.eca_finish(ah, bh, part, w1d, 3, slope), ops.eca_finish(a32, b32, part, w1d, 3, slope))
    t32, th = _rounded(torch.randn(B, H, W, 16, generator=g).to(dev))
    wt = (torch.randn(1, 16, 3, 3, generator=g) * 0.2).to(dev)
    assert maxabs(ops.tail(th, wt, slope), ops.tail(t32, wt, slope)) == 0.0                    # fp32 output from identical values


@pytest.mark.parametrize("twin_only", [True, False])
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 37, 53), (1, 480, 640)])
def test_stem_writes_an_fp16_twin(shape, twin_only, monkeypatch):
    """fp16 storage: the stem's map as IEEE fp16, written by the stem kernel itself.  Round 6 (CONFIG["gf_in_f16"], the default): ONLY that
    map -- the guided filter reads it -- and the guide still formed from the fp32 values; with the switch off: the fp32 map and its twin."""
    B, H, W = shape
    dev = _dev()
    ir, _, _ = S.make_batch(B, H, W)
    g = torch.Generator().manual_seed(8)
    w, slope = (torch.randn(32, 1, 3, 3, generator=g) * 0.4).to(dev), torch.tensor([0.2], device=dev)
    ref, gref = ops.stem(t(ir).to(dev), w, slope)
    monkeypatch.setitem(ops.CONFIG, "gf_in_f16", twin_only)
    ops.set_storage("f16")
    with ops.bf16_activations():
        feat, guide = ops.stem(t(ir).to(dev), w, slope)
        twin = ops.cast_storage(feat, True)
        if twin_only:
            assert twin is feat and feat.dtype == F16
        else:
            assert ops._TWINS[feat.data_ptr()][1] is twin           # written by the stem kernel, not by a cast pass
            assert torch.equal(feat, ref)
    assert torch.equal(guide, gref)
    assert twin.dtype == F16 and torch.equal(twin, ref.to(F16))


@pytest.mark.parametrize("shape", [(1, 24, 32), (2, 64, 96), (1, 37, 53), (1, 480, 640), (3, 100, 201)])
@pytest.mark.parametrize("engine", ["mfma2", "valu"])
def test_guided_filter_reads_the_fp16_map(shape, engine, monkeypatch):
    """paif_guided_filter_fused_fwd_hf16_y16 (round 6): y read as IEEE fp16.  Same result as the fp32-input entry on the map
    y16.float() (fp16 -> fp32 is exact, the kernels run the same arithmetic): bit-equal fp16 outputs, both engines; odd widths too."""
    B, H, W = shape
    dev = _dev()
    monkeypatch.setenv("PAIF_GF_ENGINE", engine)
    y = torch.from_numpy(S.make_smooth_feature(11, B, 32, H, W)).permute(0, 2, 3, 1).contiguous().to(dev)
    y16 = y.to(F16)
    guide = ops.channel_residue(y)
    a = ops.guided_filter_pair(guide, y16.float(), out_bf16=F16)
    b = ops.guided_filter_pair(guide, y16, out_bf16=F16)
    torch.cuda.synchronize()
    assert b.dtype == F16 and tuple(b.shape) == (2, B, H, W, 32)
    assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        ops.guided_filter_pair(guide, y16)                            # fp32 LF output from an fp16 map is not built


@pytest.mark.parametrize("shape", [(1, 24, 32), (2, 64, 96), (1, 37, 53), (1, 480, 640), (3, 100, 200)])
@pytest.mark.parametrize("engine", ["mfma2", "valu"])
def test_guided_filter_writes_high_frequency_maps_as_fp16(shape, engine, monkeypatch):
    """paif_guided_filter_fused_fwd_hf16: HF_e = y - LF_e (core/model_fusion_auto.py:531-532) rounded ONCE to fp16, from the fp32 y and
    the fp32 LF of the same engine: within half an fp16 ulp of y - LF(fp32 output) + the engines' own fp32 noise on LF (4e-6).  Both the
    matrix-core engine (in-lane 9-row delay line of y) and the all-VALU kernel it falls back to."""
    B, H, W = shape
    dev = _dev()
    monkeypatch.setenv("PAIF_GF_ENGINE", engine)
    g = torch.Generator().manual_seed(31 + H)
    y = torch.from_numpy(S.make_smooth_feature(7, B, 32, H, W)).permute(0, 2, 3, 1).contiguous().to(dev)
    guide = ops.channel_residue(y)
    lf = ops.guided_filter_pair(guide, y)
    hf = ops.guided_filter_pair(guide, y, out_bf16=F16)
    torch.cuda.synchronize()
    assert hf.dtype == F16 and tuple(hf.shape) == (2, B, H, W, 32)
    ref = y.unsqueeze(0) - lf
    err = (hf.float() - ref).abs()
    assert bool((err <= H_EPS * ref.abs() * 1.01 + 1e-5).all()), (float(err.max()), float(ref.abs().max()))
    # the point of storing HF: its magnitude, hence its fp16 rounding, is far below LF's
    assert float(ref.abs().mean()) < 0.5 * float(lf.abs().mean())


@pytest.mark.parametrize("y16", [False, True])
def test_guided_filter_12_wave_build_is_the_8_wave_build_bit_for_bit(y16, monkeypatch):
    """csrc/gf_mfma2_w12.hip (round 6: 96-column strips, three waves per SIMD; what the fp16 forward runs) against the 8-wave build of the
    same source (PAIF_GF_NW=8) on the fp16 high-frequency outputs: the two walk different strips in different runs of rows with a different
    register / LDS allocation (y delay line and own fragments through LDS, (A, b) fragments published an iteration early, half-strip plane
    loaders), but every output is formed by the same operations in the same order -- bit-identical, on ragged widths (one to three strips
    of either width, rests of every size), odd widths, heights of one piece and of several, batches whose runs cross image boundaries, and
    repeated calls (a race between waves would not repeat)."""
    dev = _dev()
    shapes = [(1, 10, 10), (1, 24, 32), (2, 33, 47), (1, 64, 96), (3, 40, 81), (1, 37, 161), (2, 100, 200), (1, 480, 640), (2, 1100, 50),
              (5, 61, 97), (1, 12, 640), (4, 480, 79)]
    for B, H, W in shapes:
        y = torch.from_numpy(S.make_smooth_feature(100 + H + W, B, 32, H, W)).permute(0, 2, 3, 1).contiguous().to(dev)
        guide = ops.channel_residue(y)
        yy = y.to(torch.float16) if y16 else y
        outs = {}
        for nw in ("8", "12", "12"):
            monkeypatch.setenv("PAIF_GF_NW", nw)
            outs.setdefault(nw, []).append(ops.guided_filter_pair(guide, yy, out_bf16=F16).clone())
        torch.cuda.synchronize()
        assert torch.equal(outs["12"][0], outs["12"][1]), (B, H, W)
        assert torch.equal(outs["8"][0], outs["12"][0]), (B, H, W, float((outs["8"][0].float() - outs["12"][0].float()).abs().max()))
    monkeypatch.setenv("PAIF_GF_NW", "10")
    with pytest.raises(RuntimeError, match="PAIF_GF_NW"):
        ops.guided_filter_pair(guide, yy, out_bf16=F16)


def _fusion_net():
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    from paif_amd.genotypes import FUSION_AT

    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    net.load_state_dict({k: t(S.formula_tensor("enhance_net." + k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()},
                        strict=True)
    return net.to(_dev())


@pytest.mark.parametrize("fused_tail", [True, False])
def test_f16_storage_overflow_guard(fused_tail, monkeypatch):
    """an fp16 map that overflows (|v| >= 65520 -> inf) must not come out as a finite, wrong fused image (the forward
    ends in tanh).  A stem weight scaled until the trunk leaves fp16's range:
    ops.check_f16_overflow() raises FloatingPointError naming set_storage("f32"); the lazy form raises from a later forward without any
    explicit call; fp32 storage of the same weights and an in-range fp16 forward raise nothing."""
    net = _fusion_net()
    ir, vis, _ = S.make_batch(1, 64, 96)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    monkeypatch.setitem(ops.CONFIG, "stem_out_fused_f32", fused_tail)
    ops.check_f16_overflow()                                   # a clean slate
    ops.set_storage("f16")
    try:
        with torch.no_grad():
            net(irt, ycc)
        ops.check_f16_overflow()                               # in range: silent
        with torch.no_grad():
            net.stem_1[0].weight.mul_(3e5)
            out = net(irt, ycc)
        del out       # (NaN here, +-1 where every overflowed sum keeps one sign: nothing in the output itself is a reliable signal)
        with pytest.raises(FloatingPointError, match="set_storage"):
            ops.check_f16_overflow()
        ops.check_f16_overflow()                               # the word was cleared
        # lazy form: the forward after the one that overflowed (or the one after that) raises on its own
        with pytest.raises(FloatingPointError, match="set_storage"):
            with torch.no_grad():
                for _ in range(4):
                    net(irt, ycc)
                    torch.cuda.synchronize()
        try:
            ops.check_f16_overflow()
        except FloatingPointError:
            pass
        ops.set_storage("f32")
        with torch.no_grad():
            net(irt, ycc)
        ops.check_f16_overflow()                               # fp32 storage has the range
    finally:
        ops.set_storage("f32")
        try:
            ops.check_f16_overflow()
        except FloatingPointError:
            pass


def test_decomposition_intermediates_forward_under_f16_storage():
    """forward(ir, vis, inter={"want_decomposition": True}) under set_storage("f16") used to raise (fp32 LF maps handed to fp16
    packs).  A forward that returns the decomposition intermediates keeps fp32 maps in every storage mode: same result as fp32 storage."""
    net = _fusion_net()
    ir, vis, _ = S.make_batch(2, 64, 96)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    a, b = {"want_decomposition": True}, {"want_decomposition": True}
    with torch.no_grad():
        ref = net(irt, ycc, inter=a)
        ops.set_storage("f16")
        try:
            out = net(irt, ycc, inter=b)
        finally:
            ops.set_storage("f32")
    assert torch.equal(out, ref) and b["lf_ir"].dtype == torch.float32 and torch.equal(b["lf_ir"], a["lf_ir"]) and torch.equal(b["fir"], a["fir"])


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 48, 64), (1, 480, 640)])
def test_fusion_forward_f16_storage_vs_fp32_storage(shape):
    """The whole inference forward in fp16 storage against the fp32-storage forward of the same network: fused plane within 2e-3 (bf16
    storage: 1.3e-2), mean 1.5e-4; the launches are the fp16 kernels (16-bit maps end to end, fp32 last map into the fp32 stem_out)."""
    B, H, W = shape
    net = _fusion_net()
    ir, vis, _ = S.make_batch(B, H, W)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    with torch.no_grad():
        ref = net(irt, ycc)
        ops.set_storage("f16")
        inter = {}
        out = net(irt, ycc, inter=inter)
        fast = net(irt, ycc)
    assert inter["ir_feature"].dtype == F16 and inter["agg"].dtype == F16 and inter["feature2"].dtype == torch.float32
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    d = (out - ref).abs()
    assert float(d.max()) <= 2.5e-3 and float(d.mean()) <= 2e-4, (float(d.max()), float(d.mean()))
    # the plain path (no intermediates requested) takes the one-kernel tail on the fp32 map: same arithmetic up to that kernel's operand split
    assert float((fast - out).abs().max()) <= 1e-4


def test_f16_storage_is_inference_only():
    """Gradient / taped passes keep fp32 storage whatever the setting says."""
    net = _fusion_net()
    ir, vis, _ = S.make_batch(1, 40, 56)
    ycc = ops.rgb2ycrcb(t(vis).to(_dev()))
    ops.set_storage("f16")
    irt = t(ir).to(_dev()).requires_grad_(True)
    net(irt, ycc).sum().backward()
    ops.set_storage("f32")
    irt2 = t(ir).to(_dev()).requires_grad_(True)
    net(irt2, ycc).sum().backward()
    assert torch.equal(irt.grad, irt2.grad)


def test_two_stream_and_graph_replay_are_bit_identical_in_f16_storage():
    net = _fusion_net()
    ir, vis, _ = S.make_batch(2, 64, 96)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    ops.set_storage("f16")
    with torch.no_grad():
        ref = net(irt, ycc)
        ops.CONFIG["two_stream"] = True
        two = net(irt, ycc)
        ops.CONFIG["two_stream"] = False
        torch.cuda.synchronize()
        gs = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(gs):
            net(irt, ycc)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=gs):
                gout = net(irt, ycc)
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(two, ref) and torch.equal(gout, ref)


NS = 32       # evaluation set of SURVEY 8(d)'s 16-bit clause: tests/golden/gq_model_b3_32x480x640.npz (oracle/make_golden_r6.py)


def _bootstrap_interval(moved, reps=20000, seed=20261004):
    """Percentile bootstrap over SAMPLES (the pixels that move come in spatial clusters: pixels are not independent draws, samples are) of
    the aggregate argmax agreement: (2.5 %, 50 %, 97.5 %)."""
    rs = np.random.RandomState(seed)
    mv = np.asarray(moved, np.float64)
    idx = rs.randint(0, len(mv), size=(reps, len(mv)))
    agg = 1.0 - mv[idx].mean(axis=1) / 307200.0
    return [float(v) for v in np.percentile(agg, [2.5, 50.0, 97.5])]


def _clause_eval(modes, golden):
    """SURVEY 8(d)'s bf16-clause numbers of the storage modes `modes` against the REFERENCE: (a) the 1x480x640 golden (sample 0: fused
    plane vs its float64 run, logits, argmax, mIoU), (b) the argmax maps / confusion matrices of the reference on THIRTY-TWO synthetic
    480x640 samples (gq_model_b3_32x480x640: samples 0-7 are the benchmarked batch; B = 1 forwards like the reference harness), with a
    bootstrap interval over samples of the aggregate agreement."""
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT
    from paif_amd.util.util import ConfusionMeter, compute_results

    g, g8, gq = golden("gf_model_b3_1x480x640"), golden("gp_model_b3_8x480x640"), golden("gq_model_b3_32x480x640")
    Hh.assert_multiclass(g["pred"], min_classes=9)
    for i in range(NS):
        Hh.assert_multiclass(gq["pred"][i], min_classes=3, min_share=0.04)
    assert np.array_equal(gq["pred"][0], g["pred"][0]) and np.array_equal(gq["pred"][:8], g8["pred"]) and gq["pred"].shape[0] == NS
    dev = _dev()
    m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
    S.load_formula_weights(m, head=Hh.HEAD480)
    m = m.to(dev)
    up = lambda x: torch.nn.functional.interpolate(x, size=(480, 640), mode="bilinear", align_corners=False).numpy()
    srt = np.sort(up(t(g["logits"])), axis=1)
    margin_ref = srt[:, -1] - srt[:, -2]
    rng = float(g["logits"].max() - g["logits"].min())
    miou = lambda conf: float(np.nanmean(compute_results(conf)[2]))
    miou_ref0, miou_ref8, miou_refq = miou(g["conf"]), miou(gq["conf"][:8].sum(0)), miou(gq["conf"].sum(0))
    report = dict(logit_range=rng, samples=NS, miou_reference_sample0=miou_ref0, miou_reference_8_samples=miou_ref8, miou_reference_32_samples=miou_refq,
                  reference_median_margin_over_range=[float(v) for v in gq["median_margin_over_range"]],
                  reference_f32_vs_f64_pixels=[int((gq["pred"][i] != gq["pred64"][i]).sum()) for i in range(NS)])
    for mode in modes:
        ops.set_storage(mode)
        moved, conf, conf8 = [], np.zeros((9, 9), np.int64), np.zeros((9, 9), np.int64)
        for i in range(NS):
            ir, vis, lab = S.make_batch(1, 480, 640, start=i)
            with torch.no_grad():
                fused, seg = m(t(ir).to(dev), t(vis).to(dev))
            meter = ConfusionMeter(9, dev)
            pred = meter.update(seg, t(lab).to(dev)).cpu().numpy()
            c = meter.conf.cpu().numpy()
            conf += c
            if i < 8:
                conf8 += c
            moved.append(int((pred[0] != gq["pred"][i]).sum()))
            if i == 0:
                d64 = (fused.cpu().double() - t(g["fused64"]).double()).abs()
                lerr = (seg.cpu() - t(g["logits"])).abs()
                dis = pred != g["pred"]
                r = dict(fused_max_abs_vs_fp64=float(d64.max()), fused_mean_abs_vs_fp64=float(d64.mean()),
                         logits_max_abs_over_range=float(lerr.max()) / rng, logits_mean_abs_over_range=float(lerr.mean()) / rng,
                         largest_reference_margin_of_a_moved_pixel_over_range=float(margin_ref[dis].max() / rng) if dis.any() else 0.0,
                         miou_delta_sample0=miou(c) - miou_ref0)
        lo, med, hi = _bootstrap_interval(moved)
        r.update(moved_pixels=moved, argmax_agreement_per_sample=[1.0 - v / 307200.0 for v in moved],
                 argmax_agreement_32_samples=1.0 - sum(moved) / (NS * 307200.0), argmax_agreement_8_samples=1.0 - sum(moved[:8]) / (8 * 307200.0),
                 argmax_agreement_bootstrap_95=[lo, hi], argmax_agreement_bootstrap_median=med,
                 samples_below_999=int(sum(1 for v in moved if 1.0 - v / 307200.0 < 0.999)),
                 argmax_agreement_sample0=1.0 - moved[0] / 307200.0, miou_delta_8_samples=miou(conf8) - miou_ref8,
                 miou_delta_32_samples=miou(conf) - miou_refq)
        report[mode] = r
    ops.set_storage("f32")
    return report


# measured on MI355X (profiles/r05_f16_storage_report.json: fused 1.8e-3 / 6.3e-5, logits 8.0e-4 / 3.9e-5 of the range); bounds = + 25-50 %
F16_CLAUSE = dict(fused_max=2.5e-3, fused_mean=1.0e-4, logits_max=1.2e-3, logits_mean=8e-5, miou=1e-3)


def test_fusion_forward_f16_storage_tolerance_clause(golden):
    """SURVEY 8(d)'s bf16-clause for the fp16 configuration, against the reference (mit_b3, calibrated head: multi-class maps with
    near-ties on every class boundary): max / mean |fused - reference| and the logit error reported and bounded; only near-tie pixels
    move; mIoU within 0.1 pt; ARGMAX AGREEMENT >= 99.9 % over THIRTY-TWO 480x640 samples (9.83 M pixels) -- asserted on the LOWER end of
    a 95 % bootstrap interval over samples (round 6; round 5 evaluated eight samples and held the mean).  On a single sample the figure
    is a noisy statistic at this level (the pixels that move are near-ties in spatial clusters): the per-sample values are reported and
    floor-bounded, not held to 99.9 % one by one (tools/storage_sensitivity.py --phase 4)."""
    report = _clause_eval(("f16", "bf16", "bf16_split", "f32"), golden)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, "f16_storage_report.json"), "w"), indent=1)
    print("storage clause report:", json.dumps(report, indent=1))
    r, lim = report["f16"], F16_CLAUSE
    assert r["fused_max_abs_vs_fp64"] <= lim["fused_max"] and r["fused_mean_abs_vs_fp64"] <= lim["fused_mean"], r
    assert r["logits_max_abs_over_range"] <= lim["logits_max"] and r["logits_mean_abs_over_range"] <= lim["logits_mean"], r
    assert r["largest_reference_margin_of_a_moved_pixel_over_range"] <= 2.0 * r["logits_max_abs_over_range"], r
    assert abs(r["miou_delta_32_samples"]) <= lim["miou"] and abs(r["miou_delta_8_samples"]) <= lim["miou"] and abs(r["miou_delta_sample0"]) <= lim["miou"], r
    assert r["argmax_agreement_32_samples"] >= 0.999, r                                                        # >= 99.9 %
    assert r["argmax_agreement_bootstrap_95"][0] >= 0.999, r                     # ... with the sampling uncertainty on the safe side
    assert min(r["argmax_agreement_per_sample"]) >= 0.9975, r                    # measured per sample 0.9982 ... 0.9997 across builds
    # the same evaluation of the other modes, for the record: fp32 storage sits at the reference's own noise; bf16 maps miss the clause
    assert report["f32"]["argmax_agreement_bootstrap_95"][0] >= 0.9998, report["f32"]
    assert report["bf16"]["argmax_agreement_bootstrap_95"][1] < 0.999, report["bf16"]


def test_fusion_f16_storage_b8_is_samplewise_the_b1_forward():
    """What bench.py times by default: B = 8 at 480x640 in fp16 storage.  Every op of the fusion network is per-sample: sample i of the
    batch is bit-equal to the B = 1 forward of that sample (persistent kernels' tile ranges straddle images only when B > 1)."""
    net = _fusion_net()
    ir, vis, _ = S.make_batch(8, 480, 640)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    ops.set_storage("f16")
    with torch.no_grad():
        full = net(irt, ycc)
        for i in (0, 3, 7):
            one = net(irt[i:i + 1].contiguous(), ycc[i:i + 1].contiguous())
            assert torch.equal(full[i:i + 1], one), i


@pytest.mark.parametrize("fmt", ["f16", "bf16"])
@pytest.mark.parametrize("shape,nres,pool", [((2, 333, 517), 0, False), ((1, 480, 640), 2, True), ((1, 37, 53), 1, False), ((3, 8, 28), 0, False),
                                             ((2, 64, 96), 2, True), ((1, 9, 29), 0, True), ((8, 480, 640), 0, False)])
def test_residual_dense_block_as_one_kernel(fmt, shape, nres, pool):
    """csrc/rdb_fused.hip: ResidualDenseBlock (operations_m.py:435-449) of the 16-bit forward as ONE launch (x1, x2 in LDS, halo recompute,
    16x16x32 MFMAs, 8 x 28 output tiles) against the three-launch form in the same storage.  Both round x1 and x2 to the storage format;
    the accumulation orders differ, so a pre-rounding value next to a rounding boundary may land on the other neighbour -- a full ulp of
    x1 / x2 on a vanishing fraction of elements, which reaches the output attenuated by the next conv's weights.  Ragged sizes, images
    smaller than a tile, tile ranges that straddle images (B = 8), extra residual maps, the fused ChannelPool."""
    from paif_amd.operations_m import ResidualDenseBlock

    B, H, W = shape
    dev = _dev()
    dt = F16 if fmt == "f16" else torch.bfloat16
    eps = H_EPS if fmt == "f16" else 2.0 ** -8
    g = torch.Generator().manual_seed(B * 1000 + H + nres)
    m = ResidualDenseBlock(32, 3, 1).eval()
    sd = m.state_dict()
    for k_, v in sd.items():
        sd[k_] = torch.randn(v.shape, generator=g) * 0.06 if v.dim() == 4 else torch.full(v.shape, 0.2)
    m.load_state_dict(sd)
    m = m.to(dev)
    mk = lambda: ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), dt)
    x, res = mk(), tuple(mk() for _ in range(nres))
    ops.set_storage(fmt)
    with ops.bf16_activations():
        c3 = torch.full((B, H, W, 4), float("nan"), device=dev)
        ref = m.forward_nhwc(x, res, None, cpool=(c3, 2) if pool else None)
        wpk = ops.rdb_fused_pack(m.conv1.conv.weight, m.conv2.conv.weight, m.conv3.conv.weight, dt)
        c1 = torch.full((B, H, W, 4), float("nan"), device=dev)
        out = ops.rdb_fused(x, wpk, m.lrelu.weight, alpha=0.333333, res=res, cpool=(c1, 2) if pool else None)
    torch.cuda.synchronize()
    assert out.dtype == dt and ref.dtype == dt and bool(torch.isfinite(out.float()).all())
    d = (out.float() - ref.float()).abs()
    scale = float(ref.float().abs().max())
    assert float(d.max()) <= 4.0 * eps * scale, (float(d.max()), scale)
    assert float((d > 1.01 * eps * ref.float().abs() + 1e-6).float().mean()) < 2e-2          # measured 0.5 %: elsewhere the same 16-bit value or its neighbour
    assert float(d.mean()) <= 0.1 * eps * float(ref.float().abs().mean()) + 1e-7
    if pool:
        assert bool(torch.isnan(c1[..., :2]).all()) and bool(torch.isfinite(c1[..., 2:]).all())
        assert float((c1[..., 2:] - c3[..., 2:]).abs().max()) <= 2.0 * eps * scale + 1e-6
    # and against the block in fp32 storage (exact-level arithmetic): the distance is the storage format's
    ops.set_storage("f32")
    ref32 = m.forward_nhwc(x.float(), tuple(r.float() for r in res), None)
    assert float((out.float() - ref32).abs().max()) <= (6.0 if fmt == "f16" else 8.0) * eps * max(1.0, float(ref32.abs().max()))


def test_fusion_forward_takes_the_one_kernel_block_at_full_size():
    """With ops.CONFIG["rdb_fused"] = True (opt-in: measured slower than the three launches) the fp16 inference forward
    at the benchmarked shape launches the three ResidualDenseBlocks of the shipped genotype as one kernel each, and the result stays
    within the storage mode's own noise of the three-launch form."""
    net = _fusion_net()
    ir, vis, _ = S.make_batch(1, 480, 640)
    irt, ycc = t(ir).to(_dev()), ops.rgb2ycrcb(t(vis).to(_dev()))
    ops.set_storage("f16")
    calls = []
    orig = ops.rdb_fused
    ops.rdb_fused = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            three = net(irt, ycc)
            assert len(calls) == 0                 # the default
            ops.CONFIG["rdb_fused"] = True
            one = net(irt, ycc)
            assert len(calls) == 3
    finally:
        ops.rdb_fused = orig
        ops.CONFIG["rdb_fused"] = False
    assert float((one - three).abs().max()) <= 1.5e-3 and float((one - three).abs().mean()) <= 6e-5
