"""GPU: the matrix-core guided filter (csrc/gf_mfma.hip) against the all-VALU kernel (PAIF_GF_ENGINE=valu) and the oracle:
ragged shapes, the f16-range fallback, timing at the bench shape."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S

dev = torch.device("cuda:0")


def run(engine, guide, x):
    os.environ["PAIF_GF_ENGINE"] = engine
    out = ops.guided_filter_pair(guide, x)
    torch.cuda.synchronize()
    return out


shapes = [(1, 24, 32), (2, 64, 96), (1, 50, 131), (3, 37, 49), (1, 480, 640), (2, 11, 10), (1, 100, 47)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:3]
for (B, H, W) in shapes:
    x = torch.from_numpy(S.make_smooth_feature(71, B, 32, H, W)).to(dev)
    xn = ops.to_nhwc(x)
    guide = ops.channel_residue(xn)
    a = run("valu", guide, xn)
    b = run("mfma", guide, xn)
    d = (a - b).abs()
    print("shape %s: max|mfma - valu| = %.3e (scale %.3f) nan %d" % ((B, H, W), float(d.max()), float(a.abs().max()), int(torch.isnan(b).sum())), flush=True)
    if float(d.max()) > 1e-5:
        idx = torch.nonzero(d > 1e-5)
        print("   first bad idx", idx[:5].tolist(), "count", idx.shape[0], "of", d.numel())
        e_, b_, y_, x_, c_ = idx[0].tolist()
        print("   valu", a[e_, b_, y_, max(0, x_ - 2):x_ + 3, c_].tolist(), "\n   mfma", b[e_, b_, y_, max(0, x_ - 2):x_ + 3, c_].tolist())
# f16-range fallback: scale the target so that 9-row sums exceed 65504
B, H, W = 1, 40, 70
x = torch.from_numpy(S.make_smooth_feature(73, B, 32, H, W)).to(dev) * 3.0e4
xn = ops.to_nhwc(x)
guide = ops.channel_residue(ops.to_nhwc(x / 3.0e4))
a = run("valu", guide, xn)
b = run("mfma", guide, xn)
print("overflow case: max|mfma - valu| = %.3e (scale %.3e)  -> fallback %s" % (float((a - b).abs().max()), float(a.abs().max()),
                                                                               "OK (bit-identical)" if torch.equal(a, b) else "MISMATCH"))
# timing at the bench shape
B, H, W = 8, 480, 640
ir, vis, _ = S.make_batch(B, H, W)
x = torch.from_numpy(S.make_smooth_feature(75, 1, 32, H, W)).to(dev).repeat(B, 1, 1, 1)
xn = ops.to_nhwc(x).contiguous()
guide = ops.channel_residue(xn)
for engine in ("valu", "mfma", "valu", "mfma"):
    os.environ["PAIF_GF_ENGINE"] = engine
    for _ in range(3):
        ops.guided_filter_pair(guide, xn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.guided_filter_pair(guide, xn)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("engine %s: %.3f ms per call (B=8 480x640; algorithmic 953.5 MB -> %.0f GB/s)" % (engine, ms, 953.5e6 / ms / 1e6), flush=True)
