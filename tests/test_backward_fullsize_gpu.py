"""GPU parity of the hand-written reverse pass AT THE SIZE THE PGD CONFIGURATION RUNS (BASELINE configs[3]: 480x640, mit_b3):
the small-shape gradient tests never reach Nk = 300 keys / N = 19,200 queries in the SR attention (64x96 has Nk = 6), the
guided-filter backward's 60-row segment seams, or the split-K / persistent dispatch of the dgrad GEMMs and convs.
Checker: the pinned CPU oracle's autograd (oracle/paif_oracle.py), run on the host beside the GPU (fp64 where the
reference's own fp32 noise floor would otherwise be the limit)."""
import numpy as np
import pytest
import torch

from paif_amd import ops, synthetic as S
from tests.helpers import t, maxabs

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _exact_convs():
    old, oldg = ops.CONFIG["conv_precision"], ops.CONFIG["gemm_precision"]
    ops.set_conv_precision("f32")
    ops.set_gemm_precision("f32")
    yield
    ops.set_conv_precision(old)
    ops.set_gemm_precision(oldg)


@pytest.mark.parametrize("B,N,Nk,C,heads", [
    (1, 19200, 300, 64, 1),     # mit_b3 stage 1 at 480x640: the longest query axis, one head
    (2, 4800, 300, 128, 2),     # stage 2
    (1, 1200, 300, 320, 5),     # stage 3 (head_dim 64, 5 heads)
    (1, 300, 300, 512, 8),      # stage 4: sr = 1, keys = queries
    (2, 777, 77, 64, 2),        # ragged: head_dim 32, N and Nk not multiples of 32, partial last workgroup
    (1, 1000, 45, 128, 2),      # ragged, head_dim 64, fewer keys than two tiles
    (1, 640, 320, 64, 1),       # 320 keys x 64 dims: K and V^T of all keys do not fit LDS in any split form -> two key chunks
])
def test_sr_attention_forward_backward_at_480x640_token_counts(B, N, Nk, C, heads):
    """core/mix_transformer.py:93-115 (softmax(q k^T * hd^-0.5) v) and its autograd, vs torch in float64: the exact-fp32 kernels
    (set_gemm_precision("f32")) and the split-bf16 forward the default arithmetic uses (products ~1e-5 relative; its
    log-sum-exp feeds the exact backward kernels)."""
    g = torch.Generator().manual_seed(N + C)
    q = torch.randn(B, N, C, generator=g)
    kv = torch.randn(B, Nk, 2 * C, generator=g)
    dout = torch.randn(B, N, C, generator=g)
    dev = _dev()
    qd, kvd = q.to(dev), kv.to(dev)
    hd = C // heads
    q64 = q.double().requires_grad_(True)
    kv64 = kv.double().requires_grad_(True)
    qh = q64.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kvh = kv64.reshape(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
    attn = ((qh @ kvh[0].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    ref = (attn @ kvh[1]).transpose(1, 2).reshape(B, N, C)
    (ref * dout.double()).sum().backward()
    prev = ops.CONFIG["gemm_precision"]
    try:
        # "auto6" = the attack loops' arithmetic: three bf16 pieces per operand, six products (2^-25) -- held to the exact kernels' bounds
        # "auto6h" = fp16 pairs (two 11-bit pieces, three MFMAs): the arithmetic the attack loops run since round 5, same bounds
        for mode, tol_out, tol_grad in (("f32", 2e-6, 2e-5), ("auto", 3e-5, 1e-4), ("auto6", 2e-6, 2e-5), ("auto6h", 2e-6, 2e-5)):
            ops.set_gemm_precision(mode)
            timer = ops.KernelTimer(lambda tag: tag.startswith("sr_attention"))
            ops.TIMER = timer
            try:
                out, lse = ops.sr_attention(qd, kvd, heads, want_lse=True)
            finally:
                ops.TIMER = None
            torch.cuda.synchronize()
            assert list(timer.summary()) == [{"f32": "sr_attention", "auto": "sr_attention_bf16x3", "auto6": "sr_attention_bf16x6", "auto6h": "sr_attention_f16x3"}[mode]]   # the entry point taken
            dq, dkv = ops.sr_attention_bwd(qd, kvd, out, dout.to(dev), lse, heads)
            assert maxabs(out.cpu().double(), ref.detach()) <= tol_out * float(ref.abs().max()), (mode, B, N, Nk, C, heads)
            for mine, r in ((dq, q64.grad), (dkv, kv64.grad)):
                assert maxabs(mine.cpu().double(), r) <= tol_grad * float(r.abs().max()), (mode, B, N, Nk, C, heads)
    finally:
        ops.set_gemm_precision(prev)


@pytest.mark.parametrize("mode,tol_out,tol_grad", [("auto", 3e-5, 1e-4), ("auto6", 2e-6, 2e-5), ("auto6h", 2e-6, 2e-5)])
def test_sr_attention_split_forms_chunk_any_key_count(mode, tol_out, tol_grad):
    """More keys than one LDS chunk holds (700 at head dim 64: three chunks with two pieces, five with three; the backward's dq kernel
    five / six): the online softmax carries over chunk boundaries, the backward's chunks are independent.  The exact kernels keep all
    keys in LDS and refuse this shape (PAIF_ENOSUP), as before."""
    B, N, Nk, C, heads = 2, 300, 700, 128, 2
    g = torch.Generator().manual_seed(77)
    q, kv, dout = torch.randn(B, N, C, generator=g), torch.randn(B, Nk, 2 * C, generator=g), torch.randn(B, N, C, generator=g)
    dev = _dev()
    hd = C // heads
    q64, kv64 = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = q64.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kvh = kv64.reshape(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (((qh @ kvh[0].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1) @ kvh[1]).transpose(1, 2).reshape(B, N, C)
    (ref * dout.double()).sum().backward()
    prev = ops.CONFIG["gemm_precision"]
    try:
        ops.set_gemm_precision(mode)
        out, lse = ops.sr_attention(q.to(dev), kv.to(dev), heads, want_lse=True)
        dq, dkv = ops.sr_attention_bwd(q.to(dev), kv.to(dev), out, dout.to(dev), lse, heads)
        assert maxabs(out.cpu().double(), ref.detach()) <= tol_out * float(ref.abs().max())
        for mine, r in ((dq, q64.grad), (dkv, kv64.grad)):
            assert maxabs(mine.cpu().double(), r) <= tol_grad * float(r.abs().max())
        ops.set_gemm_precision("f32")
        with pytest.raises(RuntimeError, match="LDS|320 keys"):
            ops.sr_attention(q.to(dev), kv.to(dev), heads)
    finally:
        ops.set_gemm_precision(prev)


def test_guided_filter_backward_at_480x640_across_segment_seams():
    """Cell_Decom.decomposition (core/model_fusion_auto.py:517-535) backward on one 480x640 32-channel map: the HIP kernels
    stream 60-row segments; the gradient (incl. the guide's arg-max / arg-min routing) must match the oracle's autograd in
    float64 at least as well as the oracle's own float32 run does."""
    from oracle import paif_oracle as O

    H, W = 480, 640
    x = t(S.make_smooth_feature(71, 1, 32, H, W))
    r = [t(S.make_feature(72 + e, (1, 32, H, W))) for e in range(2)]
    dev = _dev()
    xn = ops.to_nhwc(x.to(dev))
    guide = ops.channel_residue(xn)
    lf, ab = ops.guided_filter_pair(guide, xn, want_ab=True)
    dlf = torch.stack([ops.to_nhwc(ri.to(dev)) for ri in r]).contiguous()
    dx = ops.guided_filter_bwd(guide, xn, ab, dlf).permute(0, 3, 1, 2).cpu()

    def oracle(dtype):
        xx = x.to(dtype).requires_grad_(True)
        res = O.get_residue(xx)
        loss = sum((O.guided_filter(res, xx, 4, eps) * ri.to(dtype)).sum() for eps, ri in zip((0.001, 0.0001), r))
        loss.backward()
        return xx.grad

    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    floor = maxabs(g32.double(), g64)                      # the reference arithmetic's own float32 error
    scale = float(g64.abs().max())
    err = maxabs(dx.double(), g64)
    assert err <= max(1.5 * floor, 1e-4 * scale), (err, floor, scale)
    # seam rows of the 60-row segments specifically
    rows = [59, 60, 119, 120, 239, 240, 419, 420]
    assert maxabs(dx[:, :, rows].double(), g64[:, :, rows]) <= max(1.5 * floor, 1e-4 * scale)


def test_one_pgd_iteration_at_480x640_mit_b3_vs_oracle():
    """One full attack_both iteration (attack/attack.py:443-512) at 1x480x640 through mit_b3, exact-fp32 convs: taped
    forward, fused upsample+CE, hand-written reverse pass through both networks, vs the oracle's autograd on the host.
    Stated tolerance: loss rel. 1e-4; sign-mismatch fraction of the gradient <= 2e-3; |delta| <= eps."""
    from oracle import paif_oracle as O
    from paif_amd.attack.attack import attack_both
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT

    m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
    S.load_formula_weights(m, head=S.head_tag("mit_b3", 1, 480, 640))     # calibrated head: a multi-class, label-correlated map
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(_dev())
    ir, vis, lab = S.make_batch(1, 480, 640)
    eps, alpha = 8 / 255., 2 / 255.
    d0i = t(S.make_delta0(5, ir.shape, eps))
    d0v = t(S.make_delta0(105, vis.shape, eps))
    trace = []
    with torch.no_grad():
        d_ir, d_vis = attack_both(m, t(vis).to(_dev()), t(ir).to(_dev()), t(lab).to(_dev()), epsilon=eps, alpha=alpha, attack_iters=1,
                                  attack_loss='l_seg', attack_way='PGD', delta0_ir=d0i, delta0_vis=d0v, trace=trace)
    otrace = []
    od_ir, od_vis = O.attack_both(lambda a, b: O.model_forward(a, b, sd, "mit_b3"), t(vis), t(ir), t(lab), d0i, d0v, epsilon=eps,
                                  alpha=alpha, attack_iters=1, attack_way="PGD", trace=otrace)
    assert abs(trace[0]["loss"] - otrace[0]["loss"]) <= 1e-4 * abs(otrace[0]["loss"])
    for mine, ref in ((trace[0]["g_ir"], otrace[0]["g_ir"]), (trace[0]["g_vis"], otrace[0]["g_vis"])):
        a, b = mine.cpu().numpy(), ref.numpy()
        assert (np.sign(a) != np.sign(b)).mean() <= 2e-3
        # magnitude parity: both sides are fp32 runs through A = cov/(var + 1e-4); the reference's own fp32-vs-fp64 floor on
        # gradients through the guided filter is ~6e-3 of the scale at 64x96 and grows with the image
        assert np.abs(a - b).max() <= 5e-2 * np.abs(b).max()
        assert np.abs(a - b).mean() <= 2e-3 * np.abs(b).max()
    for mine, ref in ((d_ir, od_ir), (d_vis, od_vis)):
        a = mine.detach().cpu().numpy()
        assert (np.abs(a - ref.numpy()) > 1e-6).mean() <= 2e-3
        assert np.abs(a).max() <= eps + 1e-7


def test_parameter_gradients_at_480x640_mit_b3_vs_oracle():
    """The training step's backward AT FULL SIZE (configs[4] shape class: 480x640, mit_b3, train mode): `_loss_coupled` forward
    (BatchNorm batch statistics over 307,200 pixels, DropPath / Dropout2d masks) and every parameter gradient, vs the oracle's
    autograd on the host (float32 against float32: tolerance = a few times the arithmetic's own floor, which is largest for the
    fusion network's parameters -- their gradients pass through the guided filter)."""
    from oracle import paif_oracle as O
    from paif_amd.core.loss import Fusionloss_grad2
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT

    m = Network_MM_Searched(32, FUSION_AT, Fusionloss_grad2(), torch.nn.CrossEntropyLoss(ignore_index=255), "mit_b3", num_classes=9)
    S.load_formula_weights(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.train().to(_dev())
    ir, vis, lab = S.make_batch(1, 480, 640)
    eps = 8 / 255.
    ir_adv = np.clip(ir + S.make_delta0(7, ir.shape, eps), 0, 1).astype(np.float32)
    vis_adv = np.clip(vis + S.make_delta0(107, vis.shape, eps), 0, 1).astype(np.float32)
    mask = np.maximum(ir, vis[:, :1]).astype(np.float32)
    d = lambda a: t(a).to(_dev())
    ops.DROP_RNG.reseed(99, rank=0, step=3)
    loss = m._loss_coupled((d(ir_adv), d(ir)), (d(vis_adv), d(vis)), d(mask), d(lab))
    loss.backward()
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    O.TRAIN = O.TrainCtx(99, rank=0, step=3)
    try:
        lo = O.loss_coupled(t(ir_adv), t(vis_adv), t(mask), t(lab), sd, "mit_b3")
        lo.backward()
    finally:
        O.TRAIN = None
    assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo))
    bad, n = [], 0
    for k, p in m.named_parameters():
        ref = sd[k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        n += 1
        scale = max(float(ref.abs().max()), 1e-12)
        err = float((p.grad.cpu() - ref).abs().max())
        rel = 5e-2 if k.startswith("enhance_net.") else 5e-3
        if p.numel() == 1:
            rel = 0.25     # a PReLU slope: ONE sum over ~10 M cancelling terms, on both sides in float32
        if err > rel * scale + 1e-7:
            bad.append((round(err / scale, 4), k))
    assert not bad, sorted(bad, reverse=True)[:12]
    assert n == 623
    for k, v in m.state_dict().items():          # running statistics after one train-mode forward
        if "running_" in k:
            assert float((v.cpu() - sd[k]).abs().max()) <= 1e-5 * max(1.0, float(sd[k].abs().max())), k


def test_sr_attention_seeded_shape_sweep():
    """24 seeded random shapes (tiny and ragged query / key counts, both head dims, several heads, batch > 1) of the spatial-
    reduction attention forward + backward in both arithmetic modes against torch in float64."""
    import random

    rnd = random.Random(20261005)
    dev = _dev()
    prev = ops.CONFIG["gemm_precision"]
    try:
        for it in range(24):
            hd = rnd.choice([32, 64])
            heads = rnd.choice([1, 2, 3, 5])
            C = hd * heads
            B = rnd.randint(1, 3)
            N = rnd.choice([1, 5, 31, 32, 33, 100, 255, 256, 257, 300, 700])
            Nk = rnd.choice([1, 7, 31, 32, 33, 64, 100, 159, 160, 161, 300, 316, 320])
            g = torch.Generator().manual_seed(8000 + it)
            q = torch.randn(B, N, C, generator=g)
            kv = torch.randn(B, Nk, 2 * C, generator=g)
            dout = torch.randn(B, N, C, generator=g)
            q64 = q.double().requires_grad_(True)
            kv64 = kv.double().requires_grad_(True)
            qh = q64.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
            kvh = kv64.reshape(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
            attn = ((qh @ kvh[0].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
            ref = (attn @ kvh[1]).transpose(1, 2).reshape(B, N, C)
            (ref * dout.double()).sum().backward()
            for mode, tol_out, tol_grad in (("f32", 3e-6, 3e-5), ("auto", 4e-5, 2e-4), ("auto6", 3e-6, 3e-5), ("auto6h", 3e-6, 3e-5)):
                ops.set_gemm_precision(mode)
                out, lse = ops.sr_attention(q.to(dev), kv.to(dev), heads, want_lse=True)
                dq, dkv = ops.sr_attention_bwd(q.to(dev), kv.to(dev), out, dout.to(dev), lse, heads)
                tag = (it, mode, B, N, Nk, C, heads)
                assert torch.isfinite(out).all() and torch.isfinite(dq).all() and torch.isfinite(dkv).all(), tag
                # scale floor 1.0 = the magnitude of the N(0,1) operands: with one key the softmax is constant and dq is exactly 0
                assert maxabs(out.cpu().double(), ref.detach()) <= tol_out * max(1.0, float(ref.detach().abs().max())), tag
                assert maxabs(dq.cpu().double(), q64.grad) <= tol_grad * max(1.0, float(q64.grad.abs().max())), tag
                assert maxabs(dkv.cpu().double(), kv64.grad) <= tol_grad * max(1.0, float(kv64.grad.abs().max())), tag
    finally:
        ops.set_gemm_precision(prev)
