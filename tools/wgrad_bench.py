"""Dense-conv weight-gradient kernel at the bench shape: wgrad_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = [torch.randn(B, H, W, 32, device=dev) for _ in range(3)]
dy = torch.randn(B, H, W, 32, device=dev)
z = torch.randn(B, H, W, 32, device=dev)
slope = torch.tensor([0.2], device=dev)
for kh, dil, nsrc, act in ((3, 1, 1, 0), (3, 1, 1, 1), (3, 1, 3, 1), (1, 1, 3, 0), (7, 1, 1, 0)):
    fn = lambda: ops.conv2d_wgrad(x[:nsrc], dy, kh, dil, z=z if act else None, act=act, prelu=slope if act == 1 else None)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * H * W * kh * kh * 32 * nsrc * 32
    print("wgrad %dx%d d%d src%d act%d: %.3f ms  (%.1f TF)" % (kh, kh, dil, nsrc, act, t, fl / t / 1e9))
