"""Reverse pass of the guided-filter pair: every form against the oracle's float64 autograd (max and rms error), next to the oracle's own
float32 autograd."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from oracle import paif_oracle as O
dev = torch.device("cuda:0")
t = torch.from_numpy
cases = [("smooth 2x64x96", t(S.make_smooth_feature(71, 2, 32, 64, 96))), ("smooth 1x240x320", t(S.make_smooth_feature(75, 1, 32, 240, 320)))]
g = torch.Generator().manual_seed(5)
cases.append(("randn 2x64x96", torch.randn(2, 32, 64, 96, generator=g)))
for cname, x in cases:
    B, C, H, W = x.shape
    dlf = [torch.randn(B, 32, H, W, generator=g) for _ in range(2)]
    xn = ops.to_nhwc(x.to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    dlfn = torch.stack([ops.to_nhwc(d.to(dev)) for d in dlf]).contiguous()
    def oracle(dtype):
        xx = x.to(dtype).requires_grad_(True)
        res = O.get_residue(xx)
        loss = sum((O.guided_filter(res, xx, 4, eps) * d.to(dtype)).sum() for eps, d in zip((0.001, 0.0001), dlf))
        loss.backward()
        return xx.grad
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    print(cname, "|grad| max %.2f; oracle fp32 vs fp64: max %.2e rms %.2e" % (g64.abs().max().item(), (g32.double() - g64).abs().max().item(), (g32.double() - g64).pow(2).mean().sqrt().item()))
    _, ab = ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")
    _, mc = ops.guided_filter_pair(guide, xn, want_ab=True, tape="mc")
    outs = {}
    os.environ["PAIF_GF_BWD"] = "v1"; outs["round-1 (ab tape)"] = ops.guided_filter_bwd(guide, xn, ab, dlfn).clone()
    os.environ["PAIF_GF_BWD"] = "v2"; outs["streaming (ab tape)"] = ops.guided_filter_bwd(guide, xn, ab, dlfn).clone()
    del os.environ["PAIF_GF_BWD"]
    outs["streaming (mc tape)"] = ops.guided_filter_bwd(guide, xn, mc, dlfn).clone()
    for name, o in outs.items():
        d = o.permute(0, 3, 1, 2).cpu().double() - g64
        print("   %-22s max %.2e rms %.2e" % (name, d.abs().max().item(), d.pow(2).mean().sqrt().item()), flush=True)
