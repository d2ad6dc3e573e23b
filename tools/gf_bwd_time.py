"""Guided-filter reverse pass at B=8 480x640: streaming form (gf_backward.hip) vs the round-1 kernels (PAIF_GF_BWD=v1), timing + max |diff|."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = torch.from_numpy(S.make_smooth_feature(75, 1, 32, H, W)).to(dev).repeat(B, 1, 1, 1)
xn = ops.to_nhwc(x).contiguous()
guide = ops.channel_residue(xn)
lf, ab = ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")
lf_mc, mc = ops.guided_filter_pair(guide, xn, want_ab=True, tape="mc")
print("taped forward: max |LF ab - LF mc| = %.3e (scale %.3e)" % ((lf - lf_mc).abs().max().item(), lf.abs().max().item()))
g = torch.Generator(device="cpu").manual_seed(3)
dlf = torch.randn(2, B, H, W, 32, generator=g).to(dev)
add = torch.randn(B, H, W, 32, generator=g).to(dev)
res = {}
def timed(fn, n=10):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
for tp in ("ab", "mc", "ab", "mc"):
    print("taped forward, tape", tp, "%.3f ms" % timed(lambda: ops.guided_filter_pair(guide, xn, want_ab=True, tape=tp))[0], flush=True)
ms, res["mc"] = timed(lambda: ops.guided_filter_bwd(guide, xn, mc, dlf, add=add))
print("mc %.3f ms per reverse pass" % ms, flush=True)
for mode in ("v1", "v2", "v1", "v2"):
    os.environ["PAIF_GF_BWD"] = mode
    for _ in range(2):
        out = ops.guided_filter_bwd(guide, xn, ab, dlf, add=add)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = ops.guided_filter_bwd(guide, xn, ab, dlf, add=add)
    e1.record()
    torch.cuda.synchronize()
    res[mode] = out.clone()
    print(mode, "%.3f ms per reverse pass" % (e0.elapsed_time(e1) / 10), flush=True)
d = (res["v1"] - res["v2"]).abs().max().item()
print("max |v1 - v2| = %.3e, max |v1 - mc| = %.3e, scale %.3e" % (d, (res["v1"] - res["mc"]).abs().max().item(), res["v1"].abs().max().item()))
