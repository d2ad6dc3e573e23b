"""How test_attack_gpu.test_bf16x3_gradients_stay_within_three_floors's statistic (max |gradient error| of the split-bf16 convs against the
reference's fp32 gradient, in units of the reference's own fp32-vs-fp64 max error 6.4e-3) moves with the guided filter's tape form."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from oracle import paif_oracle as O
from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gc_fusion_2x64x96.npz")))
t = torch.from_numpy
for prec in ("bf16x3", "f32"):
    for mode in ("v1", "v2"):
        os.environ["PAIF_GF_BWD"] = mode
        ops.set_conv_precision(prec)
        net = Network_Fusion_Searched(32, None, O.FUSION_AT).eval()
        S.load_formula_weights(net)
        net.to(dev)
        ir, vis, _ = S.make_batch(2, 64, 96)
        ycc = O.rgb2ycrcb(t(vis))
        irt = t(ir).to(dev).requires_grad_(True)
        yt = ycc[:, 0:1].contiguous().to(dev).requires_grad_(True)
        fused = net(irt, yt)
        (fused * t(S.make_feature(31, tuple(fused.shape))).to(dev)).sum().backward()
        for name, mine, ref in (("d_ir", irt.grad, g["d_ir"]), ("d_y", yt.grad, g["d_y"])):
            e = (mine.cpu() - t(ref)).abs().flatten()
            q = torch.quantile(e, torch.tensor([0.5, 0.99, 0.999, 0.9999]))
            print(prec, "tape", "ab" if mode == "v1" else "mc", name, "max %.2f floors" % (e.max().item() / 6.4e-3), "median/99/99.9/99.99 %% %s floors" % ["%.3f" % (x / 6.4e-3) for x in q.tolist()],
                  "> 1 floor: %d, > 3 floors: %d of %d; |grad| max %.2f" % ((e > 6.4e-3).sum().item(), (e > 3 * 6.4e-3).sum().item(), e.numel(), t(ref).abs().max().item()), flush=True)
        print(prec, "fused max err %.2e" % (fused.detach().cpu() - t(g["fused"])).abs().max().item())
