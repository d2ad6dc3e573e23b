"""debug: which input offset does each tap of the dilation-2 LDS-DMA conv read?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from paif_amd import ops
dev = torch.device("cuda:0")
F16 = torch.float16
B, H, W = 2, 336, 512
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
base = ((yy % 32) * 32 + (xx % 32) + 1).float() / 2048.0           # positive, fp16-exact (10-bit integers)
x = torch.zeros(B, H, W, 32)
for c in range(32):
    x[:, :, :, c] = base * (1 + c % 2)
x = x.to(dev)
xh = x.to(F16)
assert torch.equal(xh.float(), x)
r = torch.zeros_like(x)
rh = r.to(F16)
ops.set_storage("f16")
for ci in (0, 5, 17, 31):
    for tap in range(9):
        w = torch.zeros(32, 32, 3, 3, device=dev)
        w[3, ci, tap // 3, tap % 3] = 1.0
        out = ops.conv2d([xh], ops.pack_conv_weight(w, 1, 32, 3, precision="f16"), 3, res=(rh,), dil=2, in_act=ops.ACT_RELU).float()[0, :, :, 3]
        best = None
        for sy in range(-5, 6):
            for sx in range(-5, 6):
                sh = torch.roll(x[0, :, :, ci], shifts=(-sy, -sx), dims=(0, 1))
                e = float((out[16:-16, 16:-16] - sh[16:-16, 16:-16]).abs().mean())
                if best is None or e < best[0]:
                    best = (e, sy, sx)
        oth = float(out.abs().mean())
        print("ci %2d tap (%d,%d): expected offset (%+d,%+d)  found (%+d,%+d) err %.4f  |  other couts max %.3f" % (
            ci, tap // 3, tap % 3, (tap // 3 - 1) * 2, (tap % 3 - 1) * 2, best[1], best[2], best[0], 0.0))
