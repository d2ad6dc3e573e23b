"""Is the PGD-10 gate's outcome for the fp16-pair reverse pass a property of the guided-filter form or of the starting point?
For N initial perturbations: the sign mismatch of the accumulated gradient between the fp16-pair arithmetic (forward and reverse pass,
GEMMs forced onto the split kernels as in tests/test_parity_default_gpu.py::test_pgd10_gate_with_the_split_gemms_forced_at_this_size)
and the exact fp32 kernels, per iteration, with the round-1 guided-filter kernels (PAIF_GF_BWD=v1) and with the streaming pair."""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from paif_amd import ops, synthetic as S
import helpers as Hh
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from paif_amd.genotypes import FUSION_AT
from paif_amd.attack.attack import attack_both
dev = torch.device("cuda:0")
EPS, ALPHA = 8 / 255., 2 / 255.
t = torch.from_numpy
m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b0", num_classes=9).eval()
S.load_formula_weights(m, head=Hh.HEAD64["mit_b0"])
m = m.to(dev)
ir, vis, lab = S.make_batch(2, 64, 96)
irt, vist, labt = t(ir).to(dev), t(vis).to(dev), t(lab).to(dev)
NSEED = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PATHS = sys.argv[2:] or ["mc"]        # ONE guided-filter path per process: a captured attack graph is reused within a process whatever PAIF_GF_BWD says later

def run(mode, seed):
    old = dict(ops.CONFIG)
    try:
        if mode == "exact":
            ops.set_conv_precision("f32"); ops.set_gemm_precision("f32"); ops.set_attack_precision("exact")
        else:
            ops.set_conv_precision("bf16x3"); ops.set_gemm_precision("auto"); ops.set_attack_precision("bf16x6")
            ops.CONFIG["gemm_split_min_m"] = 1
            ops.CONFIG["attack_fwd_f16x3"] = True
            ops.CONFIG["attack_bwd_f16x3"] = mode == "f16x3_fb"
        trace = []
        d0i, d0v = t(S.make_delta0(seed, ir.shape, EPS)).to(dev), t(S.make_delta0(100 + seed, vis.shape, EPS)).to(dev)
        with torch.no_grad():
            attack_both(m, vist, irt, labt, epsilon=EPS, alpha=ALPHA, attack_iters=10, attack_loss='l_seg', attack_way='PGD', delta0_ir=d0i, delta0_vis=d0v, trace=trace)
        return [(np.sign(s["g_ir"].cpu().numpy()), np.sign(s["g_vis"].cpu().numpy())) for s in trace]
    finally:
        ops.CONFIG.clear(); ops.CONFIG.update(old)

out = {}
for path in PATHS:
    if path == "v1":
        os.environ["PAIF_GF_BWD"] = "v1"
    else:
        os.environ.pop("PAIF_GF_BWD", None)
    for mode in ("f16x3_f", "f16x3_fb"):
        rows = []
        for seed in range(NSEED):
            ex = run("exact", seed)
            ap = run(mode, seed)
            mm = [max(float((a[0] != e[0]).mean()), float((a[1] != e[1]).mean())) for a, e in zip(ap, ex)]
            rows.append(mm)
            print(path, mode, "seed", seed, "sign mismatch vs the exact kernels x1e4 per iteration:", [round(x * 1e4, 1) for x in mm], flush=True)
        r = np.array(rows)
        out["%s/%s" % (path, mode)] = {"per_seed_per_iteration": r.tolist(), "runs_with_a_mismatch_by_iteration_10": int((r.max(1) > 0).sum()),
                                       "mean_at_iteration_10": float(r[:, -1].mean()), "max_at_iteration_10": float(r[:, -1].max())}
        print(path, mode, "runs with any mismatch: %d of %d; at iteration 10: mean %.1e max %.1e" % ((r.max(1) > 0).sum(), NSEED, r[:, -1].mean(), r[:, -1].max()), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pgd_seed_study_%s.json" % "_".join(PATHS)), "w"), indent=1)
