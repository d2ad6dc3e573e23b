"""HIP-event time of the stem kernel (3x3 conv 1->32 + PReLU + guide, fp32 map + bf16 twin) at B=8 480x640 with the loaded library
(PAIF_LIB= variants built with -DPAIF_STEM_GRID=n): python tools/stem_time.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(0)
img = torch.rand(B, 1, H, W, generator=g).to(dev)
w = (torch.randn(32, 1, 3, 3, generator=g) * 0.3).to(dev)
slope = torch.tensor([0.2], device=dev)
res = []
for twin in (False, True):
    ops._ACT_BF16[0] = twin
    for _ in range(3):
        ops.stem(img, w, slope)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.stem(img, w, slope)
    e1.record()
    torch.cuda.synchronize()
    ops._TWINS.clear()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = B * H * W * (128 + 4 + (64 if twin else 0)) / 1e6
    res.append("%s %6.1f us %5.2f TB/s" % ("fp32 + bf16 twin" if twin else "fp32 map", us, mb / us))
ops._ACT_BF16[0] = False
print(os.path.basename(os.environ.get("PAIF_LIB", "default")), " | ".join(res))
