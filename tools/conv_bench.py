"""Micro-benchmark of the dense conv kernel at the bench shape (B=8, 480x640, C=32) + a copy baseline.
   PAIF_CONV_WS=1 selects the wave-specialised kernel.  Prints us per launch and algorithmic TB/s."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = [torch.randn(B, H, W, 32, device=dev) for _ in range(3)]
r = [torch.randn(B, H, W, 32, device=dev) for _ in range(3)]
out = torch.empty(B, H, W, 32, device=dev)
MAP = B * H * W * 32 * 4 / 1e9

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

t = timeit(lambda: out.copy_(x[0]))
print("copy 1r+1w            %7.1f us  %.2f TB/s" % (t, 2 * MAP / t * 1e6 / 1e3))
t = timeit(lambda: torch.add(x[0], x[1], out=out))
print("add  2r+1w            %7.1f us  %.2f TB/s" % (t, 3 * MAP / t * 1e6 / 1e3))
t = timeit(lambda: x[0].sum())
print("sum  1r               %7.1f us  %.2f TB/s" % (t, 1 * MAP / t * 1e6 / 1e3))
for kh, dil in ((1, 1), (3, 1), (3, 2)):
    for nsrc in (1, 2, 3):
        for nres in (0, 3):
            w = torch.randn(32, 32 * nsrc, kh, kh, device=dev) * 0.05
            wpk = ops.pack_conv_weight(w, nsrc, 32, kh)
            fn = lambda: ops.conv2d(x[:nsrc], wpk, kh, dil=dil, res=tuple(r[:nres]), out=out)
            t = timeit(fn)
            maps = nsrc + 1 + nres
            print("conv%dx%d d%d src%d res%d  %7.1f us  %.2f TB/s" % (kh, kh, dil, nsrc, nres, t, maps * MAP / t * 1e6 / 1e3))
