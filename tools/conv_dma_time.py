"""Times the 3x3 bf16 conv forms of the fusion forward (B=8 480x640) with the loaded library: python tools/conv_dma_time.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(0)
ops.set_storage("bf16")
maps = [ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev) * 0.5), True) for _ in range(6)]
out = []
for nsrc, nres in [(1, 0), (1, 1), (2, 0), (3, 1), (3, 3)]:
    w = (torch.randn(32, 32 * nsrc, 3, 3, generator=g) * 0.05).to(dev)
    wpk = ops.pack_conv_weight(w, nsrc, 32, 3, precision="bf16x3")
    srcs, res = maps[:nsrc], tuple(maps[3:3 + nres])
    for _ in range(3):
        ops.conv2d(srcs, wpk, 3, res=res, act=1, prelu=torch.tensor([0.2], device=dev))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    slope = torch.tensor([0.2], device=dev)
    e0.record()
    for _ in range(reps):
        ops.conv2d(srcs, wpk, 3, res=res, act=1, prelu=slope)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = B * H * W * 64 * (nsrc + nres + 1) / 1e6
    out.append("(%d,%d) %6.1f us %5.2f TB/s" % (nsrc, nres, us, mb / us))
print(os.environ.get("PAIF_LIB", "default").split("_")[-1], " | ".join(out))
