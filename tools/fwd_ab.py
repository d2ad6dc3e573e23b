"""A/B of one ops.CONFIG switch on the bf16 fusion forward (B=8 480x640), same process, alternating: python tools/fwd_ab.py <key> [reps]
e.g. stem_out_fused, dilconv_dense"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
from paif_amd.genotypes import FUSION_AT
key = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
S.load_formula_weights(net)
net = net.to(dev)
ir, vis, _ = S.make_batch(8, 480, 640)
ir, vis = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)
ops.set_storage("bf16")
with torch.no_grad():
    ycc = ops.rgb2ycrcb(vis)


def run():
    with torch.no_grad():
        for _ in range(5):
            net(ir, ycc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(ir, ycc)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    ops.CONFIG[key] = False
    ref = net(ir, ycc).clone()
    ops.CONFIG[key] = True
    got = net(ir, ycc).clone()
torch.cuda.synchronize()
print("%s: max |on - off| = %.3e" % (key, float((ref - got).abs().max())))
for rnd in range(3):
    ops.CONFIG[key] = False
    off = run()
    ops.CONFIG[key] = True
    on = run()
    print("%s: off %.3f ms | on %.3f ms (%+.1f %%)" % (key, off, on, 100 * (off / on - 1)))
