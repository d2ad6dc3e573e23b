"""Guided filter at 480x640: a batch of 8 copies against the B = 1 run, bit for bit (taped forward, tape, reverse pass, inference forward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
dev = torch.device("cuda:0")
H, W = 480, 640
x = torch.from_numpy(S.make_smooth_feature(75, 1, 32, H, W))
g = torch.Generator().manual_seed(3)
dlf = torch.randn(2, 1, H, W, 32, generator=g)
add = torch.randn(1, H, W, 32, generator=g)
outs = []
for B in (1, 8):
    xn = ops.to_nhwc(x.repeat(B, 1, 1, 1).to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    lf, tape = ops.guided_filter_pair(guide, xn, want_ab=True)
    dy = ops.guided_filter_bwd(guide, xn, tape, dlf.repeat(1, B, 1, 1, 1).to(dev).contiguous(), add=add.repeat(B, 1, 1, 1).to(dev).contiguous())
    hf = ops.guided_filter_pair(guide, xn, out_bf16=torch.float16)
    outs.append((lf, tape.mc, dy, hf))
names = ("lf", "mc", "dy", "hf16")
for b in range(8):
    print(b, [(n, bool(torch.equal(o8[:, b] if n != "dy" else o8[b], o1[:, 0] if n != "dy" else o1[0]))) for n, o1, o8 in zip(names, outs[0], outs[1])])
