"""Run one conv config N times (for rocprofv3 counter passes): conv_one.py kh dil nsrc nres [reps]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
kh, dil, nsrc, nres = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = [torch.randn(B, H, W, 32, device=dev) for _ in range(nsrc)]
r = [torch.randn(B, H, W, 32, device=dev) for _ in range(nres)]
out = torch.empty(B, H, W, 32, device=dev)
w = torch.randn(32, 32 * nsrc, kh, kh, device=dev) * 0.05
wpk = ops.pack_conv_weight(w, nsrc, 32, kh)
for _ in range(reps):
    ops.conv2d(x, wpk, kh, dil=dil, res=tuple(r), out=out)
torch.cuda.synchronize()
