"""GPU: PGD-10 trajectory (2x64x96, mit_b0) against the reference's float64 run for every combination of conv / GEMM arithmetic:
where does the default arithmetic's sign-mismatch come from?  Prints one line per combination."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from paif_amd.attack.attack import attack_both
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from paif_amd.genotypes import FUSION_AT

g = dict(np.load("tests/golden/gn_attack_PGD10.npz"))
dev = torch.device("cuda:0")
m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b0", num_classes=9).eval()
S.load_formula_weights(m, head=S.head_tag("mit_b0", 2, 64, 96))      # the calibrated head of the golden (multi-class maps)
m = m.to(dev)
ir, vis, lab = S.make_batch(2, 64, 96)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
ops.set_attack_precision("fast")      # the loop runs whatever the two settings below say
convs = sys.argv[1].split(",") if len(sys.argv) > 1 else ("f32", "bf16x6", "bf16x3")
gemms = sys.argv[2].split(",") if len(sys.argv) > 2 else ("f32", "auto6", "bf16x6", "auto", "bf16x3")
for conv in convs:
    for gemm in gemms:
        ops.set_conv_precision(conv)
        ops.set_gemm_precision(gemm)
        trace = []
        with torch.no_grad():
            d_ir, d_vis = attack_both(m, t(vis).to(dev), t(ir).to(dev), t(lab).to(dev), epsilon=8 / 255., alpha=2 / 255., attack_iters=10,
                                      attack_loss='l_seg', attack_way='PGD', delta0_ir=t(g["d0_ir"]), delta0_vis=t(g["d0_vis"]), trace=trace)
        sm = [max(float((np.sign(s["g_ir"].cpu().numpy()) != g["sign64_ir_per_iter"][i]).mean()),
                  float((np.sign(s["g_vis"].cpu().numpy()) != g["sign64_vis_per_iter"][i]).mean())) for i, s in enumerate(trace)]
        lr = [abs(s["loss"] - g["losses64"][i]) / abs(g["losses64"][i]) for i, s in enumerate(trace)]
        dm = max(float((np.abs(d_ir.detach().cpu().numpy() - g["delta64_ir"]) > 1e-6).mean()), float((np.abs(d_vis.detach().cpu().numpy() - g["delta64_vis"]) > 1e-6).mean()))
        print("conv %-7s gemm %-7s sign-mismatch vs fp64 it1 %.2e it3 %.2e it5 %.2e it10 %.2e max %.2e | loss rel max %.2e | differing delta %.2e" % (
            conv, gemm, sm[0], sm[2], sm[4], sm[9], max(sm), max(lr), dm), flush=True)
