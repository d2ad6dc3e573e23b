"""HIP-event time of the small per-pixel kernels of the bf16 fusion forward at B=8 480x640 (spa_blend, tail, channel_pool2, eca_finish)
with the loaded library (PAIF_LIB= for A/B builds): python tools/elementwise_time.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(0)
ops.set_storage("bf16")
a = ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), True)
b = ops.cast_storage(ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev)), True)
t16 = torch.randn(B, H, W, 16, generator=g).to(dev).to(torch.bfloat16)
w5 = (torch.randn(1, 4, 5, 5, generator=g) * 0.2).to(dev)
wt = (torch.randn(1, 16, 3, 3, generator=g) * 0.2).to(dev)
slope = torch.tensor([0.25], device=dev)
comp = ops.channel_pool2(a, b)
part = torch.rand(ops.lib().paif_conv2d_blocks(B, H, W), 32, device=dev)
w1d = torch.randn(3, device=dev)


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


print(os.path.basename(os.environ.get("PAIF_LIB", "default")),
      "spa_blend %.1f us | tail %.1f us | channel_pool2 %.1f us | eca_finish %.1f us" % (
          timeit(lambda: ops.spa_blend(comp, w5, a, b)), timeit(lambda: ops.tail(t16, wt, slope)),
          timeit(lambda: ops.channel_pool2(a, b)), timeit(lambda: ops.eca_finish(a, b, part, w1d, 3, slope))))
