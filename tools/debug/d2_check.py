"""debug: the dilation-2 LDS-DMA conv against the fp32-storage conv: error by tile row / column"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from paif_amd import ops
dev = torch.device("cuda:0")
F16 = torch.float16
B, H, W = 2, 333, 517
for nres in (1, 3):
    g = torch.Generator().manual_seed(5 + nres)
    x32 = ops.cast_storage(ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), F16), torch.float32)
    r32 = [ops.cast_storage(ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), F16), torch.float32) for _ in range(nres)]
    xh, rh = ops.cast_storage(x32, F16), [ops.cast_storage(r, F16) for r in r32]
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).to(dev).to(F16).float()
    kw = dict(dil=2, in_act=ops.ACT_RELU)
    ref = ops.conv2d([x32], ops.pack_conv_weight(w, 1, 32, 3, precision="bf16x3"), 3, res=tuple(r32), **kw)
    ops.set_storage("f16")
    out = ops.conv2d([xh], ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3, res=tuple(rh), **kw).float()
    ops.set_storage("f32")
    # also without residuals' contribution
    err = (out - ref).abs()
    print("nres", nres, "max err", float(err.max()), "mean", float(err.mean()), "ref absmax", float(ref.abs().max()))
    e = err[0].amax(-1)            # [H, W]
    rows = [float(e[r::8].mean()) for r in range(8)]
    print(" err by row mod 8:", ["%.3f" % v for v in rows])
    cols = [float(e[:, c::32].mean()) for c in range(0, 32, 4)]
    print(" err by col mod 32 (every 4th):", ["%.3f" % v for v in cols])
    # is out equal to the conv WITHOUT relu? or with dilation 1?
    for name, kw2 in (("no relu", dict(dil=2)), ("dil1 relu", dict(dil=1, in_act=ops.ACT_RELU)), ("dil1", dict(dil=1))):
        alt = ops.conv2d([x32], ops.pack_conv_weight(w, 1, 32, 3, precision="bf16x3"), 3, res=tuple(r32), **kw2)
        print("  vs", name, float((out - alt).abs().mean()))
