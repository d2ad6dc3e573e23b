"""GPU tests of the three-piece split arithmetic ("bf16x6", PAIF_CONV_BF16X6 / GEMM precision 3): every operand as bf16 hi + mid + lo
(the fp32 value to 2^-27), six bf16 MFMAs per product (the three dropped cross terms are 2^-27 relative), fp32 accumulate -- the
arithmetic of the attack loops (ops.CONFIG["attack_precision"] = "bf16x6").  It must sit as close to float64 as the exact fp32-MFMA
kernels do; the trajectory-level statement (PGD-10 sign mismatch vs the reference's float64 run) is in test_parity_default_gpu.py."""
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
    ops.CONFIG.update(old)


@pytest.mark.parametrize("kh,dil,nsrc,cout", [(3, 1, 1, 32), (3, 1, 3, 32), (1, 1, 2, 32), (7, 1, 1, 32), (3, 2, 1, 32), (5, 1, 1, 32), (3, 1, 1, 16)])
def test_conv_bf16x6_is_fp32_level(kh, dil, nsrc, cout):
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
    err = {}
    for prec in ("f32", "bf16x6"):
        wpk = ops.pack_conv_weight(w, nsrc, 32, kh, precision=prec)
        out = ops.conv2d(xs, wpk, kh, dil=dil, cout=cout, scale=scale, shift=shift, act=ops.ACT_PRELU, prelu=slope, alpha=0.5, res=(res,) if res is not None else ())
        err[prec] = float((ops.to_nchw_view(out).double() - ref).abs().max() / ref.abs().max())
    assert err["bf16x6"] <= max(2.0 * err["f32"], 5e-7), err


def test_conv_bf16x6_gradient_hooks_match_exact_kernel():
    """The dgrad staging / epilogue hooks (activation-derivative transforms, saved pre-activation) in the three-piece arithmetic."""
    dev = _dev()
    B, H, W = 2, 50, 77
    g = torch.Generator().manual_seed(5)
    x = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev))
    aux = ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev))
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).to(dev)
    sc = (torch.rand(32, generator=g) + 0.5).to(dev)
    slope = torch.tensor([0.25], device=dev)
    outs = {}
    for prec in ("f32", "bf16x6"):
        wpk = ops.pack_conv_weight(w, 1, 32, 3, precision=prec)
        o1 = ops.conv2d([x], wpk, 3, in_act=ops.IN_DPRELU, in_aux=aux, in_prelu=slope, in_scale=sc, in_alpha=0.5)
        o2, z = ops.conv2d([x], wpk, 3, act=ops.ACT_PRELU, prelu=slope, want_aux=True)
        o3 = ops.conv2d([x], wpk, 3, epi_dact=1, epi_aux=aux, prelu=slope)
        outs[prec] = (o1, o2, z, o3)
    for a, b in zip(outs["f32"], outs["bf16x6"]):
        assert float((a - b).abs().max()) <= 3e-6 * float(a.abs().max())


@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (2500, 64, 256), (19200, 1280, 320), (4800, 512, 2048)])
def test_gemm_bf16x6_is_fp32_level(M, N, K):
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ w.double().t() + bias.double()
    err = {}
    for prec in ("f32", "bf16x6"):
        ops.set_gemm_precision(prec)
        out = ops.gemm(a, w, shift=bias)
        err[prec] = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err["bf16x6"] <= max(2.0 * err["f32"], 5e-7), err
