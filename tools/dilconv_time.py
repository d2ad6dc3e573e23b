"""Times DilConv (operations_m.py:494-506) at B=8 480x640 in the bf16 inference forward: depthwise + 1x1 (two kernels) against
the one dense dilated conv with the composed weight, for 0..2 extra residual maps: python tools/dilconv_time.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
from paif_amd.operations_m import DilConv
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(0)
F32 = os.environ.get("DIL_F32") == "1"
ops.set_storage("f32" if F32 else "bf16")
maps = [ops.to_nhwc(torch.randn(B, 32, H, W, generator=g).to(dev) * 0.5) for _ in range(3)]
if not F32:
    maps = [ops.cast_storage(m_, True) for m_ in maps]
KEY = "dilconv_dense_f32" if F32 else "dilconv_dense"
m = DilConv(32, 32, 3, 2).eval().to(dev)


def run(nres):
    for _ in range(3):
        with torch.no_grad():
            m.forward_nhwc(maps[0], res=tuple(maps[1:1 + nres]))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(reps):
            m.forward_nhwc(maps[0], res=tuple(maps[1:1 + nres]))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for nres in (0, 1, 2):
    ops.CONFIG[KEY] = False
    two = run(nres)
    ops.CONFIG[KEY] = True
    one = run(nres)
    print("extra residual maps %d: depthwise + 1x1 %6.1f us | one dense conv %6.1f us (PAIF_CONV_WS=%s)" % (nres, two, one, os.environ.get("PAIF_CONV_WS", "1")))
