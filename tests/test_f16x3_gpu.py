"""GPU tests of the fp16-pair arithmetic ("f16x3", PAIF_CONV_F16X3; round 5): every operand as TWO IEEE fp16 pieces (hi = rn(v),
lo = rn(v - hi): 22 significant bits), three fp16 MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulate; the weight side carries an
exact 2^8 so that its lo pieces stay normal fp16 numbers, undone on the accumulators.  It is the arithmetic of the FORWARD passes of the
attack loops (ops.attack_forward_arithmetic, CONFIG["attack_fwd_f16x3"]): half the MFMAs of the three-piece bf16 form.  It must sit as
close to float64 as the exact fp32-MFMA kernels do on the data of this path; the trajectory-level statement (PGD-10 sign mismatch vs the
reference's float64 run, with the split GEMMs forced at the small size) is tests/test_parity_default_gpu.py."""
import ctypes

import pytest
import torch

from paif_amd import ops

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore():
    old = dict(ops.CONFIG)
    yield
    ops.CONFIG.clear()
    ops.CONFIG.update(old)


@pytest.mark.parametrize("kh,dil,nsrc,cout", [(3, 1, 1, 32), (3, 1, 3, 32), (1, 1, 2, 32), (7, 1, 1, 32), (3, 2, 1, 32), (5, 1, 1, 32), (3, 1, 1, 16)])
def test_conv_f16x3_is_fp32_level(kh, dil, nsrc, cout):
    """The full epilogue (affine, PReLU, alpha, residual, saved pre-activation) on ragged maps, against float64; bound = the bf16x6 test's."""
    dev = _dev()
    B, H, W = 2, 70, 101
    g = torch.Generator().manual_seed(kh * 10 + dil + nsrc)
    xs = [ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)) for _ in range(nsrc)]
    w = (torch.randn(cout, 32 * nsrc, kh, kh, generator=g) * 0.05).to(dev)
    res = ops.to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev)) if cout == 32 else None
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    slope = torch.tensor([0.2], device=dev)
    x64 = torch.cat([ops.to_nchw_view(x) for x in xs], 1).double()
    z = torch.nn.functional.conv2d(x64, w.double(), padding=dil * (kh - 1) // 2, dilation=dil) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref = torch.where(z >= 0, z, z * 0.2) * 0.5
    if res is not None:
        ref = ref + ops.to_nchw_view(res).double()
    err, aux_err = {}, {}
    for prec in ("f32", "f16x3"):
        wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision=prec)
        out, aux = ops.conv2d(xs, wpk, kh, dil=dil, cout=cout, scale=scale, shift=shift, act=ops.ACT_PRELU, prelu=slope, alpha=0.5,
                              res=(res,) if res is not None else (), want_aux=True)
        err[prec] = float((ops.to_nchw_view(out).double() - ref).abs().max() / ref.abs().max())
        aux_err[prec] = float((ops.to_nchw_view(aux).double() - z).abs().max() / z.abs().max())      # the taped forward's pre-activation
    assert err["f16x3"] <= max(2.0 * err["f32"], 5e-7), err
    assert aux_err["f16x3"] <= max(2.0 * aux_err["f32"], 5e-7), aux_err


@pytest.mark.parametrize("scale_x,scale_w,bound", [(1.0, 0.05, 1.0), (30.0, 0.3, 1.0), (0.02, 0.05, 4.0), (1.0, 2e-4, 4.0)])
def test_conv_f16x3_over_the_magnitudes_of_the_path(scale_x, scale_w, bound):
    """fp16's narrow exponent: activations below 0.125 and weights below 5e-4 have SUBNORMAL lo pieces (absolute error <= 3e-8 per
    operand) -- the relative error of an output may then exceed the exact kernel's, by a small factor (bound), never the split-bf16 level."""
    dev = _dev()
    g = torch.Generator().manual_seed(int(scale_x * 1000) + int(scale_w * 1e6))
    x = ops.to_nhwc((torch.randn(2, 32, 48, 64, generator=g) * scale_x).to(dev))
    w = (torch.randn(32, 32, 3, 3, generator=g) * scale_w).to(dev)
    ref = torch.nn.functional.conv2d(ops.to_nchw_view(x).double(), w.double(), padding=1)
    err = {}
    for prec in ("f32", "f16x3", "bf16x3"):
        out = ops.conv2d([x], ops.pack_conv_weight(w, 1, 32, 3, precision=prec), 3)
        err[prec] = float((ops.to_nchw_view(out).double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err["f16x3"] <= bound * max(err["f32"], 3e-7), err
    assert err["f16x3"] <= 0.5 * err["bf16x3"], err


def test_conv_f16x3_gradient_hooks_match_exact_kernel_and_16_bit_maps_are_refused():
    """The dgrad staging / epilogue modes (activation-derivative transforms) in the fp16-pair arithmetic, on gradients scaled to O(1)
    the way the attack loop does it; fp16-STORED maps are another configuration (precision "f16") and refused."""
    dev = _dev()
    B, H, W = 2, 50, 77
    g = torch.Generator().manual_seed(5)
    dy = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev))
    aux = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev))
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).to(dev)
    slope = torch.tensor([0.25], device=dev)
    outs = {}
    for prec in ("f32", "f16x3"):
        wpk = ops.pack_conv_weight(w, 1, 32, 3, precision=prec)
        o1 = ops.conv2d([dy], wpk, 3, in_act=ops.IN_DPRELU, in_prelu=slope, in_aux=aux)
        o2 = ops.conv2d([dy], wpk, 3, in_act=ops.IN_DRELU, in_aux=aux, epi_aux=aux, epi_dact=2)
        outs[prec] = (o1, o2)
    for a, b in zip(outs["f32"], outs["f16x3"]):
        assert float((a - b).abs().max() / a.abs().max()) <= 1e-6
    wpk = ops.pack_conv_weight(w, 1, 32, 3, precision="f16x3")
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.conv2d([dy.half()], wpk, 3)
    d = ops._lib.ConvDesc()                                  # the C ABI refuses 16-bit storage itself
    out = torch.empty_like(dy)
    d.src[0], d.nsrc, d.cin, d.wpk, d.kh, d.dil, d.precision = ops._p(dy), 1, 32, ops._p(wpk.data), 3, 1, 6
    d.out, d.cout, d.alpha, d.storage = ops._p(out), 32, 1.0, 3
    assert ops.lib().paif_conv2d_fwd(ctypes.byref(d), B, H, W, None) != 0
    assert "f16x3" in ops.lib().paif_last_error().decode() or "fp16" in ops.lib().paif_last_error().decode()


@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (3000, 1280, 320), (2500, 64, 4096), (2049, 72, 256), (100, 512, 512)])
def test_gemm_f16x3_is_fp32_level(M, N, K):
    """The GEMM form (plain and split-K, bias + GELU + residual epilogue) against float64; the dgrad prologue keeps the three-piece form."""
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double()) + res.double()
    err = {}
    for prec in ("f32", "bf16x6", "f16x3"):
        ops.set_gemm_precision(prec)
        timer = ops.KernelTimer(lambda tag: tag.startswith("gemm_mfma"))
        ops.TIMER = timer
        try:
            y = ops.gemm(a, w, shift=bias, act=ops.ACT_GELU, res=res)
        finally:
            ops.TIMER = None
        torch.cuda.synchronize()
        assert list(timer.summary()) == ["gemm_mfma_" + prec]
        err[prec] = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err["f16x3"] <= max(2.0 * err["f32"], 1e-6), err
    # the dgrad prologue (ReLU mask, per-column scale) in the same arithmetic
    mask = (torch.randn(M, K, generator=g) > 0).float().to(dev)
    colscale = (torch.rand(K, generator=g) + 0.5).to(dev)
    ref2 = (a.double() * mask.double() * colscale.double()) @ w.double().t()
    ops.set_gemm_precision("f16x3")
    y2 = ops.gemm(a, w, a_mask=mask, a_scale=colscale)
    ops.set_gemm_precision("f32")
    y2e = ops.gemm(a, w, a_mask=mask, a_scale=colscale)
    e2, e2e = float((y2.double() - ref2).abs().max() / ref2.abs().max()), float((y2e.double() - ref2).abs().max() / ref2.abs().max())
    assert e2 <= max(2.0 * e2e, 1e-6), (e2, e2e)


def test_attack_forward_arithmetic_switches_only_the_forward():
    """The context the attack loop puts around forward_taped: conv packs "f16x3", GEMM rule "auto6h"; outside it (the reverse pass) the
    three-piece forms; nothing changes outside an attack loop or with the switch off."""
    assert ops.CONFIG["attack_fwd_f16x3"] is True                      # the product default
    ops.CONFIG["attack_bwd_f16x3"] = False
    seen = []
    with ops.attack_arithmetic():
        seen.append((ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]))
        with ops.attack_forward_arithmetic():
            seen.append((ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"], ops.pack_precision()))
        seen.append((ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]))
    assert seen == [("bf16x6", "auto6"), ("f16x3", "auto6h", "f16x3"), ("bf16x6", "auto6")]
    with ops.attack_forward_arithmetic():                                # not inside an attack loop: untouched
        assert (ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]) == ("bf16x3", "auto")
    ops.CONFIG["attack_fwd_f16x3"] = False
    with ops.attack_arithmetic(), ops.attack_forward_arithmetic():
        assert (ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]) == ("bf16x6", "auto6")
    ops.set_attack_precision("exact")
    ops.CONFIG["attack_fwd_f16x3"] = True
    with ops.attack_arithmetic(), ops.attack_forward_arithmetic():
        assert (ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]) == ("f32", "f32")


def test_attack_loop_reports_an_fp16_range_overflow_instead_of_returning_garbage():
    """A gradient scale far too large (2^60 instead of ~2^9 at this size) pushes the scaled gradients past 65504: the loop raises and
    names the switches; with the product scale the same call returns a finite perturbation inside the eps-ball."""
    from paif_amd import synthetic as S
    from paif_amd.attack.attack import attack_both
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT
    from tests import helpers as Hh
    from tests.helpers import t

    dev = _dev()
    m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b0", num_classes=9).eval()
    S.load_formula_weights(m, head=Hh.HEAD64["mit_b0"])
    m = m.to(dev)
    ir, vis, lab = S.make_batch(2, 64, 96)
    args = (m, t(vis).to(dev), t(ir).to(dev), t(lab).to(dev))
    kw = dict(epsilon=8 / 255., alpha=2 / 255., attack_iters=2, attack_loss="l_seg", attack_way="PGD",
              delta0_ir=t(S.make_delta0(1, ir.shape, 8 / 255.)).to(dev), delta0_vis=t(S.make_delta0(2, vis.shape, 8 / 255.)).to(dev))
    with torch.no_grad():
        d_ir, d_vis = attack_both(*args, **kw)
    assert bool(torch.isfinite(d_ir).all()) and float(d_ir.abs().max()) <= 8 / 255. + 1e-7
    ops.CONFIG["attack_grad_scale_log2"] = 60
    with torch.no_grad(), pytest.raises(FloatingPointError, match="fp16's exponent range"):
        attack_both(*args, **kw)
    ops.CONFIG["attack_grad_scale_log2"] = None
    ops.CONFIG.update(attack_fwd_f16x3=False, attack_bwd_f16x3=False, attn_f16x3=False)      # the way out the message names
    with torch.no_grad():
        d2_ir, _ = attack_both(*args, **kw)
    assert float((d2_ir != d_ir).float().mean()) <= 2e-3                       # same trajectory up to near-zero gradient elements


@pytest.mark.parametrize("M,N,K", [(19200, 320, 320), (4097, 64, 256), (2400, 512, 2048), (1000, 9, 256), (5000, 40, 72)])
def test_gemm_wgrad_f16x3_is_fp32_level_on_gradient_magnitudes(M, N, K):
    """dW = dY^T X and db = sum dY of a Linear layer (core/mix_transformer.py:22-25) in the fp16-pair arithmetic the training step's
    reverse pass uses (ops.wgrad_scale): dY of the magnitude a pixel-averaged loss produces (1e-7 ... 1e-5, with a long tail), scaled by
    a power of two inside the kernel; against float64 and against the exact fp32-MFMA kernel (ragged N / K, accumulation into .grad)."""
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    dy = (torch.randn(M, N, generator=g) * 3e-7 * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    refw, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    w_e, b_e = ops.gemm_wgrad(dy, x)                                   # outside the context: the exact kernel
    with ops.wgrad_scale(8 * 480 * 640):
        w_h, b_h = ops.gemm_wgrad(dy, x)
        acc_w, acc_b = torch.ones(N, K, device=dev), torch.ones(N, device=dev)
        ops.gemm_wgrad(dy, x, out_w=acc_w, out_b=acc_b)                # accumulating form
    assert ops.CONFIG["_wgrad_scale"] is None
    e_e = float((w_e.double() - refw).abs().max() / refw.abs().max())
    e_h = float((w_h.double() - refw).abs().max() / refw.abs().max())
    assert e_h <= max(3.0 * e_e, 2e-6), (e_h, e_e)
    assert not torch.equal(w_h, w_e)                                    # (it is the other kernel)
    assert float((b_h.double() - refb).abs().max() / refb.abs().max()) <= 2e-6
    assert float((acc_w - 1.0 - w_h).abs().max()) <= 1e-6 * float(w_h.abs().max()) + 1e-7 and float((acc_b - 1.0 - b_h).abs().max()) <= 1e-7
