"""stem_out of the bf16 forward at B=8 480x640: conv 32->16 (LDS-DMA kernel) + tail against the one-kernel form: python tools/stem_out_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(0)
ops.set_storage("bf16")
x = ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), True)
w1 = (torch.randn(16, 32, 3, 3, generator=g) * 0.08).to(dev)
w2 = (torch.randn(1, 16, 3, 3, generator=g) * 0.2).to(dev)
slope = torch.tensor([0.3], device=dev)
w0 = ops.pack_conv_weight(w1, 1, 32, 3)
wso = ops.stem_out_pack(w1, w2)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


two = timeit(lambda: ops.tail(ops.conv2d([x], w0, 3, 1, cout=16), w2, slope))
one = timeit(lambda: ops.stem_out_fused(x, wso, slope))
a = ops.tail(ops.conv2d([x], w0, 3, 1, cout=16), w2, slope)
b = ops.stem_out_fused(x, wso, slope)
print("conv 32->16 + tail %.1f us | one kernel (+ ring) %.1f us | max |difference| %.2e (the two-kernel form rounds the 16-channel map to bf16)" % (two, one, float((a - b).abs().max())))
