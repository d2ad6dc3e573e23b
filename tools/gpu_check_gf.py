"""Fused guided filter vs the two-kernel form: max diff per shape + timing at the bench shape."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
dev = torch.device("cuda:0")
for shape in [(1, 24, 32), (2, 50, 70), (1, 130, 97), (1, 10, 10), (1, 251, 49), (1, 480, 640)]:
    B, H, W = shape
    y = ops.to_nhwc(torch.from_numpy(S.make_smooth_feature(5, B, 32, H, W)).to(dev))
    g = ops.channel_residue(y)
    f = ops.guided_filter_pair(g, y)
    t, ab = ops.guided_filter_pair(g, y, want_ab=True)
    d = (f - t).abs()
    i = d.flatten().argmax().item()
    idx = []
    for n in reversed(d.shape):
        idx.append(i % n); i //= n
    print(shape, "maxdiff %.3e" % d.max().item(), "scale %.3f" % t.abs().max().item(), "at", idx[::-1], "ab max %.1f" % ab.abs().max().item())
y = torch.randn(8, 480, 640, 32, device=dev); g = ops.channel_residue(y)
for name, kw in (("fused", {}), ("two", {"want_ab": True})):
    for _ in range(3): ops.guided_filter_pair(g, y, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.guided_filter_pair(g, y, **kw)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.3f ms" % (e0.elapsed_time(e1) / 10))
