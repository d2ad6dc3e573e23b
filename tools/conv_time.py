"""Time one dense-conv configuration at the bench shape: conv_time.py kh dil nsrc nres  -> us per launch (HIP events, 20 reps)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
for spec in sys.argv[1:]:
    kh, dil, nsrc, nres = (int(v) for v in spec.split(","))
    x = [torch.randn(B, H, W, 32, device=dev) for _ in range(nsrc)]
    r = [torch.randn(B, H, W, 32, device=dev) for _ in range(nres)]
    out = torch.empty(B, H, W, 32, device=dev)
    wpk = ops.pack_conv_weight(torch.randn(32, 32 * nsrc, kh, kh, device=dev) * 0.05, nsrc, 32, kh)
    fn = lambda: ops.conv2d(x, wpk, kh, dil=dil, res=tuple(r), out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print("conv%dx%d d%d src%d res%d  %7.1f us   %.0f TF algorithmic" % (kh, kh, dil, nsrc, nres, t, 2 * B * H * W * kh * kh * 32 * nsrc * 32 / t / 1e6))
