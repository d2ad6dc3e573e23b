import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from paif_amd import ops, synthetic as S
import helpers as Hh
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from paif_amd.genotypes import FUSION_AT
from paif_amd.attack.attack import attack_both
dev = torch.device("cuda:0")
EPS, ALPHA = 8 / 255., 2 / 255.
t = torch.from_numpy
m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b0", num_classes=9).eval()
S.load_formula_weights(m, head=Hh.HEAD64["mit_b0"]); m = m.to(dev)
ir, vis, lab = S.make_batch(2, 64, 96)
irt, vist, labt = t(ir).to(dev), t(vis).to(dev), t(lab).to(dev)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9
def run(mode, path):
    if path == "v1": os.environ["PAIF_GF_BWD"] = "v1"
    else: os.environ.pop("PAIF_GF_BWD", None)
    old = dict(ops.CONFIG)
    try:
        if mode == "exact":
            ops.set_conv_precision("f32"); ops.set_gemm_precision("f32"); ops.set_attack_precision("exact")
        else:
            ops.set_conv_precision("bf16x3"); ops.set_gemm_precision("auto"); ops.set_attack_precision("bf16x6")
            ops.CONFIG["gemm_split_min_m"] = 1; ops.CONFIG["attack_fwd_f16x3"] = True; ops.CONFIG["attack_bwd_f16x3"] = True
        trace = []
        d0i, d0v = t(S.make_delta0(seed, ir.shape, EPS)).to(dev), t(S.make_delta0(100 + seed, vis.shape, EPS)).to(dev)
        with torch.no_grad():
            attack_both(m, vist, irt, labt, epsilon=EPS, alpha=ALPHA, attack_iters=1, attack_loss='l_seg', attack_way='PGD', delta0_ir=d0i, delta0_vis=d0v, trace=trace)
        return trace[0]["g_ir"].cpu().numpy().copy(), trace[0]["g_vis"].cpu().numpy().copy()
    finally:
        ops.CONFIG.clear(); ops.CONFIG.update(old)
def run3(mode, path, tape=None):
    if tape: os.environ["PAIF_GF_TAPE"] = tape
    else: os.environ.pop("PAIF_GF_TAPE", None)
    return run(mode, path)
combos = [("exact", "v1", None), ("exact", "mc", None), ("exact", "v2", "ab"), ("f16", "v1", None), ("f16", "mc", None), ("f16", "v2", "ab")]
res = {}
for rep in range(3):
    for c in combos:
        res.setdefault(c, []).append(run3(*c))
base = res[("exact", "v1", None)][0]
for c in combos:
    r = res[c]
    rep_diff = max(np.abs(r[0][w] - r[i][w]).max() for i in (1, 2) for w in (0, 1))
    print(c, "repeat-to-repeat max |diff| %.3e; vs exact/v1: ir %.3e vis %.3e" % (rep_diff, np.abs(r[0][0] - base[0]).max(), np.abs(r[0][1] - base[1]).max()))
