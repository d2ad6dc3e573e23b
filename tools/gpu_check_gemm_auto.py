"""Parity of the "auto" GEMM arithmetic (split-bf16 where K >= 256) against the exact-fp32 GEMMs on the mit_b3 goldens."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from paif_amd.genotypes import FUSION_AT
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gf_model_b3_1x480x640.npz")))
m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
S.load_formula_weights(m); m.cuda()
ir, vis, _ = S.make_batch(1, 480, 640)
for mode in ("f32", "auto", "bf16x3"):
    ops.set_gemm_precision(mode)
    with torch.no_grad():
        fused, seg = m(torch.from_numpy(ir).cuda(), torch.from_numpy(vis).cuda())
    rng = float(g["logits"].max() - g["logits"].min())
    e32 = float(np.abs(seg.cpu().numpy() - g["logits"]).max()); e64 = float(np.abs(seg.cpu().numpy() - g["logits64"]).max())
    floor = float(np.abs(g["logits"] - g["logits64"]).max())
    up = torch.nn.functional.interpolate(seg.cpu(), size=(480, 640), mode="bilinear", align_corners=False)
    print("%-7s logits: max|d| vs ref fp32 %.2e, vs ref fp64 %.2e (reference's own fp32 floor %.2e), range %.2f, argmax agreement %.5f" % (
        mode, e32, e64, floor, rng, float((up.argmax(1).numpy() == g["pred"]).mean())))
