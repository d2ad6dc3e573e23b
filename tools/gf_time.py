import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = torch.from_numpy(S.make_smooth_feature(75, 1, 32, H, W)).to(dev).repeat(B, 1, 1, 1)
xn = ops.to_nhwc(x).contiguous()
guide = ops.channel_residue(xn)
for _ in range(3):
    ops.guided_filter_pair(guide, xn)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.guided_filter_pair(guide, xn)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("PAIF_LIB", "default"), os.environ.get("PAIF_GF_ENGINE", "mfma2"), "fp32 out %.3f ms" % (e0.elapsed_time(e1) / 20))
e0.record()
for _ in range(20):
    ops.guided_filter_pair(guide, xn, out_bf16=True)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("PAIF_LIB", "default"), os.environ.get("PAIF_GF_ENGINE", "mfma2"), "bf16 out %.3f ms" % (e0.elapsed_time(e1) / 20))
