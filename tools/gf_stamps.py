"""Cycle stamps of one wave of the round-4 guided-filter kernel (library built with -DGF2_STAMP=<block id> [-DGF2_STAMP_WAVE=w]):
PAIF_LIB=paif_amd/lib/libpaif_hip_stamp.so python tools/gf_stamps.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S, _lib
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = torch.from_numpy(S.make_smooth_feature(75, 1, 32, H, W)).to(dev).repeat(B, 1, 1, 1)
xn = ops.to_nhwc(x).contiguous()
guide = ops.channel_residue(xn)
L = ops.lib()
lf = torch.empty((2, B, H, W, 32), device=dev, dtype=torch.float32)
ws = torch.zeros(L.paif_guided_filter_fused_workspace_floats(B, H, W), device=dev, dtype=torch.float32)
for _ in range(3):
    _lib.check(L.paif_guided_filter_fused_fwd(ops._p(guide), ops._p(xn), ops._p(lf), 1e-3, 1e-4, ops._p(ws), B, H, W, ops._stream()), "gf")
torch.cuda.synchronize()
npix = B * H * W
st = ws[4 * npix + 2:4 * npix + 2 + 60].cpu().numpy().view(np.uint64).reshape(5, 6).astype(np.int64)
names = ["top", "split done", "published", "barrier out", "stage2 done", "stage1 done"]
for it in range(4):
    t0 = st[it, 0]
    print("it %2d:" % (30 + it), " ".join("%s +%d" % (names[j], st[it, j] - t0) for j in range(1, 6)), "| next top +%d" % (st[it + 1, 0] - t0))
