"""LF maps of every guided-filter form against the oracle's float64 filter, on the stem features of the 2x64x96 test batch
(the input test_bf16x3_gradients_stay_within_three_floors runs on) and on a smooth 480x640 map."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from oracle import paif_oracle as O
from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
dev = torch.device("cuda:0")
t = torch.from_numpy
net = Network_Fusion_Searched(32, None, O.FUSION_AT).eval()
S.load_formula_weights(net)
sd = {k: v.clone() for k, v in net.state_dict().items()}
ir, vis, _ = S.make_batch(2, 64, 96)
ycc = O.rgb2ycrcb(t(vis))
feats = {}
with torch.no_grad():
    # the stems' outputs through the oracle (fp32): what the guided filter reads
    import torch.nn.functional as F
    for name, img, key in (("ir", t(ir), "stem_ir"), ("vis", ycc[:, 0:1], "stem_vis")):
        w = [k for k in sd if k.startswith(key) and k.endswith("0.weight")]
        feats[name] = None
cases = []
x1 = t(S.make_smooth_feature(71, 2, 32, 64, 96)); cases.append(("smooth 2x64x96", x1))
x2 = t(S.make_smooth_feature(75, 1, 32, 480, 640)); cases.append(("smooth 1x480x640", x2))
g = torch.Generator().manual_seed(5); cases.append(("randn 2x64x96", torch.randn(2, 32, 64, 96, generator=g)))
for cname, x in cases:
    xn = ops.to_nhwc(x.to(dev)).contiguous()
    guide = ops.channel_residue(xn)
    x64 = x.double()
    res = O.get_residue(x64)
    ref = [O.guided_filter(res, x64, 4, eps) for eps in (0.001, 0.0001)]
    ref32 = [O.guided_filter(O.get_residue(x), x, 4, eps) for eps in (0.001, 0.0001)]
    forms = {
        "round-1 pair (ab)": ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")[0],
        "streaming pair (mc)": ops.guided_filter_pair(guide, xn, want_ab=True, tape="mc")[0],
        "fused, matrix cores": ops.guided_filter_pair(guide, xn),
    }
    os.environ["PAIF_GF_ENGINE"] = "valu"
    forms["fused, all-VALU"] = ops.guided_filter_pair(guide, xn).clone()
    del os.environ["PAIF_GF_ENGINE"]
    for e in range(2):
        print(cname, "eps", (0.001, 0.0001)[e], "oracle fp32 vs fp64: max %.2e rms %.2e" % ((ref32[e].double() - ref[e]).abs().max().item(), (ref32[e].double() - ref[e]).pow(2).mean().sqrt().item()))
        for name, lf in forms.items():
            d = (lf[e].permute(0, 3, 1, 2).cpu().double() - ref[e])
            print("   %-22s max %.2e rms %.2e" % (name, d.abs().max().item(), d.pow(2).mean().sqrt().item()), flush=True)
