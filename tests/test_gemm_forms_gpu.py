"""GPU tests of the two round-5 GEMM forms against the GEMM they replace (csrc/gemm_mfma.hip), which the seeded shape sweeps of
tests/test_seg_gpu.py pin against fp64:

* `ops.conv_gemm` -- the strided convs of MiT (OverlapPatchEmbed.proj core/mix_transformer.py:168-169, Attention.sr :74) with the im2col
  matrix gathered inside the GEMM's loader (`paif_gemm_conv_fwd`): BIT-IDENTICAL to `ops.im2col` + `ops.gemm` (same k order, same
  arithmetic), on the mit_b3 tap / channel combinations, ragged maps (padding on every side), the split-K path and both split
  arithmetics; refuses what it is not built for.
* `csrc/gemm_split2.hip` (opt-in, `ops.CONFIG["gemm2"]`) -- the wide-tile split-bf16 GEMM with pre-split weights: bit-identical to
  `paif_gemm_fwd` for every tile width and both arithmetics, ragged M, epilogue (bias, GELU, residual) included; the packed-weight
  cache follows in-place weight updates."""
import ctypes

import numpy as np
import pytest
import torch

from paif_amd import ops

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class _cfg:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: ops.CONFIG[k] for k in self.kw}
        ops.CONFIG.update(self.kw)

    def __exit__(self, *exc):
        ops.CONFIG.update(self.old)


CONV_CASES = [  # (B, H, W, Cin, N, k, stride, pad)
    (8, 64, 96, 64, 64, 8, 8, 0),        # Attention.sr, stage 1 (sr_ratio 8): K = 4096, split-K
    (8, 32, 48, 128, 128, 4, 4, 0),      # stage 2 (sr_ratio 4)
    (9, 30, 40, 320, 320, 2, 2, 0),      # stage 3 (sr_ratio 2), the 480x640 token grid
    (2, 61, 83, 64, 128, 3, 2, 1),       # OverlapPatchEmbed 2 on a ragged map: padding rows / columns on every side, M % 128 != 0
    (3, 30, 40, 128, 320, 3, 2, 1),      # OverlapPatchEmbed 3
    (16, 15, 20, 320, 512, 3, 2, 1),     # OverlapPatchEmbed 4 at B = 16: M = 1280 tokens... below the M >= 2048 rule in "auto"
    (1, 17, 23, 32, 40, 5, 3, 2),        # not a mit shape: 5 x 5 taps, stride 3, N not a multiple of 64
]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x6"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_gemm_gather_bit_identical_to_im2col_gemm(case, prec):
    B, H, W, Cin, N, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 100000)
    x = torch.randn(B, H, W, Cin, generator=g).to(_dev())
    w = (torch.randn(N, Cin, k, k, generator=g) * 0.05).to(_dev())
    bias = torch.randn(N, generator=g).to(_dev())
    wp = ops.pack_conv_gemm_weight(w)
    with _cfg(gemm_precision=prec, gemm_gather=False):
        ref, OH, OW = ops.conv_gemm(x, wp, k, s, p, shift=bias)
    with _cfg(gemm_precision=prec, gemm_gather=True):
        got, OH2, OW2 = ops.conv_gemm(x, wp, k, s, p, shift=bias)
    assert (OH, OW) == (OH2, OW2) and got.shape == ref.shape == (B, OH * OW, N)
    assert torch.equal(got, ref)
    # and against the definition (torch conv on the host, fp64): the pair itself is pinned elsewhere, this guards the test's own plumbing
    want = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), bias.cpu().double(), stride=s, padding=p)
    want = want.permute(0, 2, 3, 1).reshape(B, OH * OW, N)
    err = float((got.cpu().double() - want).abs().max() / want.abs().max())
    assert err <= (3e-5 if prec == "bf16x3" else 2e-6), err


@pytest.mark.parametrize("case", [(2, 64, 96, 3, 64, 7, 4, 3), (1, 61, 83, 3, 32, 7, 4, 3), (3, 33, 47, 3, 64, 5, 2, 2), (2, 20, 28, 5, 24, 3, 2, 1),
                                  (16, 120, 160, 3, 64, 7, 4, 3)], ids=lambda c: "x".join(map(str, c)))
def test_conv_gemm_exact_gather_of_the_3_channel_patch_embed(case):
    """OverlapPatchEmbed 1 (core/mix_transformer.py:168: Conv2d(3, 64, 7, stride 4, padding 3)) in the exact fp32 MFMA: the gathered form
    decodes the columns through a table (147 taps padded to 160) -- bit-identical to im2col + gemm, ragged maps and other small forms."""
    B, H, W, Cin, N, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g).to(_dev())
    w = (torch.randn(N, Cin, k, k, generator=g) * 0.05).to(_dev())
    bias = torch.randn(N, generator=g).to(_dev())
    wp = ops.pack_conv_gemm_weight(w)
    assert wp.shape[1] <= 160
    for prec in ("f32", "auto", "auto6"):                  # "auto" / "auto6" resolve to the exact MFMA at K < 256
        with _cfg(gemm_precision=prec, gemm_gather=False):
            ref, OH, OW = ops.conv_gemm(x, wp, k, s, p, shift=bias)
        with _cfg(gemm_precision=prec, gemm_gather=True):
            timer = ops.KernelTimer(lambda tag: True)
            ops.TIMER = timer
            try:
                got, _, _ = ops.conv_gemm(x, wp, k, s, p, shift=bias)
            finally:
                ops.TIMER = None
            torch.cuda.synchronize()
        assert torch.equal(got, ref), prec
        assert list(timer.summary()) == ["gemm_mfma_f32"]       # one launch: no im2col, no second GEMM
    want = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), bias.cpu().double(), stride=s, padding=p)
    want = want.permute(0, 2, 3, 1).reshape(B, OH * OW, N)
    assert float((got.cpu().double() - want).abs().max() / want.abs().max()) <= 2e-6


def test_conv_gemm_keeps_the_pair_where_no_gathered_form_is_built_and_refuses_bad_calls():
    dev = _dev()
    x = torch.randn(8, 16, 24, 64, device=dev)
    w = torch.randn(64, 64, 2, 2, device=dev) * 0.05
    wp = ops.pack_conv_gemm_weight(w)
    with _cfg(gemm_precision="f32"):                       # exact fp32 MFMA at K = 256 > 160: the im2col pair
        y32, _, _ = ops.conv_gemm(x, wp, 2, 2, 0)
        col = ops.im2col(x, 2, 2, 0, wp.shape[1])
        assert torch.equal(y32, ops.gemm(col.view(8, -1, wp.shape[1]), wp))
    L = ops.lib()
    err = lambda: L.paif_last_error().decode()
    out = torch.empty(8 * 8 * 12, 64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.paif_gemm_conv_fwd(p(x), 8, 16, 24, 64, 2, 2, 0, p(wp), None, None, 0, None, 0, p(out), 64, 64, 0, 1, None, None)
    assert rc != 0 and "K <= 160" in err()
    rc = L.paif_gemm_conv_fwd(p(x), 8, 16, 24, 64, 2, 2, 0, p(wp), None, None, 0, None, 0, p(out), 64, 64, 2, 1, None, None)
    assert rc != 0 and "precision" in err()
    x3 = torch.randn(2, 16, 24, 3, device=dev)
    rc = L.paif_gemm_conv_fwd(p(x3), 2, 16, 24, 3, 7, 4, 3, p(wp), None, None, 0, None, 0, p(out), 64, 64, 1, 1, None, None)
    assert rc != 0 and "Cin" in err()
    rc = L.paif_gemm_conv_fwd(p(x), 8, 16, 24, 64, 2, 2, 0, p(wp), None, None, 0, None, 0, p(out), 64, 64, 1, 3, None, None)
    assert rc != 0 and "splits" in err()     # 8 k tiles do not split in 3; and no workspace


G2_CASES = [  # (M, N, K): nt = 5, 4, 2, 1 by paif_gemm2_plan
    (1000, 320, 320), (777, 1280, 96), (300, 256, 512), (513, 1024, 64), (4100, 128, 160), (129, 384, 32), (2050, 64, 256), (64, 192, 1280),
]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x6"])
@pytest.mark.parametrize("shape", G2_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wide_tile_gemm_bit_identical_to_gemm(shape, prec):
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 7 * N + 13 * K)
    a = torch.randn(M, K, generator=g).to(_dev())
    w = (torch.randn(N, K, generator=g) * 0.05).to(_dev())
    bias, res = torch.randn(N, generator=g).to(_dev()), torch.randn(M, N, generator=g).to(_dev())
    nt = ops.lib().paif_gemm2_plan(M, N, K, 1 if prec == "bf16x3" else 3)
    want_nt = 5 if N % 320 == 0 else 4 if N % 256 == 0 else 2 if N % 128 == 0 else 1
    assert nt == (want_nt if not (prec == "bf16x6" and want_nt == 5) else 0)     # three pieces x 320 rows do not fit LDS
    for act, r in ((ops.ACT_NONE, None), (ops.ACT_GELU, res), (ops.ACT_RELU, res)):
        with _cfg(gemm_precision=prec, gemm2=False):
            ref = ops.gemm(a, w, shift=bias, act=act, res=r)
        for force in (True, 1, 2):
            with _cfg(gemm_precision=prec, gemm2=force):
                got = ops.gemm(a, w, shift=bias, act=act, res=r)
            assert torch.equal(got, ref), (force, act)


def test_wide_tile_gemm_pack_follows_weight_updates():
    dev = _dev()
    a = torch.randn(512, 128, device=dev)
    w = torch.nn.Parameter(torch.randn(256, 128, device=dev) * 0.05)
    with _cfg(gemm_precision="bf16x3", gemm2=True):
        y0 = ops.gemm(a, w)
        with torch.no_grad():
            w.mul_(2.0)                                    # in place: same object, same address, new version
        y1 = ops.gemm(a, w)
        with torch.no_grad():
            w.data = w.data * 0.5                          # replaced storage
        y2 = ops.gemm(a, w)
    assert torch.equal(y1, y0 * 2.0) and torch.equal(y2, y0)
    assert len(ops._GEMM2_PACKS) >= 1
    del w
    import gc
    gc.collect()


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16x6", "f16x3"])
@pytest.mark.parametrize("case", [(2, 64, 96, 64, 8), (3, 32, 48, 128, 4), (9, 30, 40, 320, 2), (1, 16, 24, 32, 2), (2, 30, 41, 64, 2)],
                         ids=lambda c: "x".join(map(str, c)))
def test_gemm_col2im_scatter_bit_identical_to_gemm_then_col2im(case, prec):
    """The input gradient of the spatial-reduction conv (kernel = stride = sr_ratio, core/mix_transformer.py:74): the col2im done by the
    dgrad GEMM's epilogue addresses (`paif_gemm_col2im_fwd`) against `ops.gemm` + `ops.col2im`; the last case (W not a multiple of sr)
    takes the pair itself."""
    B, H, W, C, sr = case
    g = torch.Generator().manual_seed(sum(case))
    M = B * ((H - sr) // sr + 1) * ((W - sr) // sr + 1)
    dy = torch.randn(M, C, generator=g).to(_dev())
    wt = (torch.randn(sr * sr * C, C, generator=g) * 0.05).to(_dev())
    with _cfg(gemm_precision=prec, gemm_gather=False):
        ref = ops.gemm_col2im(dy, wt, B, H, W, C, sr)
    with _cfg(gemm_precision=prec, gemm_gather=True):
        got = ops.gemm_col2im(dy, wt, B, H, W, C, sr)
    assert got.shape == ref.shape == (B, H, W, C) and torch.equal(got, ref)
    if H % sr == 0 and W % sr == 0:          # and the definition: the transposed conv of the gradient, float64 on the host
        want = torch.nn.functional.conv_transpose2d(dy.cpu().double().view(B, H // sr, W // sr, C).permute(0, 3, 1, 2),
                                                    wt.cpu().double().view(sr, sr, C, C).permute(3, 2, 0, 1), stride=sr).permute(0, 2, 3, 1)
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err <= (3e-5 if prec == "bf16x3" else 3e-6), err
