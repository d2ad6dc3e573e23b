"""TEST INFRASTRUCTURE -- golden vectors added in round 3.  Run ONLY in the build container:   python oracle/make_golden_r3.py

Same rules as oracle/make_golden.py: imports the real reference from /root/reference (oracle/ref_import.py + oracle/shims),
formula weights and inputs (paif_amd/synthetic.py), stores the REFERENCE's outputs as small data fixtures.

gn_attack_PGD10: the reference's own `attack_both` (attack/attack.py:417-514) run for TEN iterations -- the count BASELINE
configs[3] / robust_test.py:42 use -- on 2 pairs of 64x96 through mit_b0, followed by the PGD-eval harness's forward on the
attacked pair (robust_test.py:143-166): per-iteration loss, the sign of the RUNNING gradient sum after every iteration
(delta.grad is never zeroed, :501-512), the final delta, the attacked prediction / confusion matrix.  The same run is repeated
with the reference in FLOAT64: the disagreement between the reference's float32 run and its float64 run (loss trajectory,
sign-mismatch fraction per iteration, differing delta elements, moved pixels) is the reference arithmetic's OWN noise floor for
this metric -- the yardstick the GPU path's default (split-bf16) arithmetic is held to in tests/test_parity_default_gpu.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import paif_oracle as O  # noqa: E402
from oracle.make_golden import build_model, t, npy, save  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402

EPS, ALPHA, ITERS = 8 / 255.0, 2 / 255.0, 10


def run_reference_attack(R, model, ir, vis, lab, d0_ir, d0_vis, dtype):
    """The reference's attack_both with the start perturbation pinned through the global RNG (it draws delta0 itself,
    attack/attack.py:434,439: `torch.zeros_like(X).uniform_(-eps, eps)`, ir first then vis) and the running gradient sum
    recorded after every iteration (read from the delta Variables through a forward hook on the model call)."""
    seg_maps, holders = [], {}

    def recording_model(a, b):
        fz, sg = model(a, b)
        seg_maps.append(sg.detach().clone())
        return fz, sg

    torch.manual_seed(1234)
    chk_ir = torch.zeros_like(ir).uniform_(-EPS, EPS)
    chk_vis = torch.zeros_like(vis).uniform_(-EPS, EPS)
    assert torch.equal(chk_ir, d0_ir.to(dtype)) and torch.equal(chk_vis, d0_vis.to(dtype))
    for p_ in model.parameters():
        p_.grad = None
    torch.manual_seed(1234)
    with torch.no_grad():
        d_ir, d_vis = R["attack"].attack_both(recording_model, vis, ir, lab, epsilon=EPS, alpha=ALPHA, attack_iters=ITERS,
                                              attack_loss="l_seg", attack_way="PGD")
    losses = []
    for sg in seg_maps:
        outp = torch.nn.functional.interpolate(sg, size=lab.shape[1:], mode="bilinear", align_corners=False)
        losses.append(float(R["attack"].Seg_loss()(outp, lab)))
    with torch.no_grad():                                       # robust_test.py:160-166: forward on the attacked pair
        fused, seg = model(ir + d_ir, vis + d_vis)
        up = torch.nn.functional.interpolate(seg, size=lab.shape[1:], mode="bilinear", align_corners=False)
        pred = up.argmax(1)
    return dict(d_ir=d_ir.detach(), d_vis=d_vis.detach(), g_ir=d_ir.grad.detach(), g_vis=d_vis.grad.detach(),
                losses=np.array(losses), fused=fused, logits=seg, pred=pred)


def per_iteration_signs(model_fwd, ir, vis, lab, d0_ir, d0_vis, dtype):
    """Running-gradient-sum signs after EVERY iteration: the oracle's restatement of the loop (pinned to the reference's final
    state below) in the requested dtype."""
    trace = []
    O.attack_both(model_fwd, vis.to(dtype), ir.to(dtype), lab, d0_ir.to(dtype), d0_vis.to(dtype), EPS, ALPHA, ITERS, "PGD", trace=trace)
    return trace


def main():
    R = ref_import.load()
    torch.set_num_threads(8)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    m0 = build_model(R, "mit_b0", head=S.head_tag("mit_b0", 2, 64, 96))
    ir, vis, lab = (t(a) for a in S.make_batch(2, 64, 96))
    torch.manual_seed(1234)
    d0_ir = torch.zeros_like(ir).uniform_(-EPS, EPS)
    d0_vis = torch.zeros_like(vis).uniform_(-EPS, EPS)

    r32 = run_reference_attack(R, m0, ir, vis, lab, d0_ir, d0_vis, torch.float32)
    # the oracle's loop must land where the reference's does (pins the per-iteration trace below to the reference)
    sd = {k: v.clone() for k, v in m0.state_dict().items()}
    tr32 = per_iteration_signs(lambda a, b: O.model_forward(a, b, sd, "mit_b0"), ir, vis, lab, d0_ir, d0_vis, torch.float32)
    assert np.allclose([s["loss"] for s in tr32], r32["losses"], rtol=1e-5), ([s["loss"] for s in tr32], r32["losses"])
    mism = float((torch.sign(tr32[-1]["g_ir"]) != torch.sign(r32["g_ir"])).float().mean())
    assert mism <= 1e-3, mism

    # float64: the oracle's loop (validated against the reference's float32 run just above) on the float64 state dict, from the
    # SAME float32 delta0 (a float64 uniform_ would draw different numbers)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    fwd64 = lambda a, b: O.model_forward(a, b, sd64, "mit_b0")
    tr64 = per_iteration_signs(fwd64, ir, vis, lab, d0_ir, d0_vis, torch.float64)
    od_ir, od_vis = O.attack_both(fwd64, vis.double(), ir.double(), lab, d0_ir.double(), d0_vis.double(), EPS, ALPHA, ITERS, "PGD")
    with torch.no_grad():
        f64, s64 = fwd64(ir.double() + od_ir, vis.double() + od_vis)
        up64 = torch.nn.functional.interpolate(s64, size=lab.shape[1:], mode="bilinear", align_corners=False)
        pred64 = up64.argmax(1)

    with torch.no_grad():                                       # the same model on the CLEAN pair: what the attack destroys
        _, seg_c = m0(ir, vis)
        pred_clean = torch.nn.functional.interpolate(seg_c, size=lab.shape[1:], mode="bilinear", align_corners=False).argmax(1)
    conf_clean = O.confusion_matrix(lab.numpy(), pred_clean.numpy())
    conf = O.confusion_matrix(lab.numpy(), r32["pred"].numpy())
    conf64 = O.confusion_matrix(lab.numpy(), pred64.numpy())
    iou = O.compute_results(conf)[2]
    iou64 = O.compute_results(conf64)[2]
    sign = lambda x: np.sign(npy(x)).astype(np.int8)
    # the reference arithmetic's own floor: float32 run vs float64 run, per iteration
    floor_sign = np.array([[float((sign(a["g_ir"]) != sign(b["g_ir"])).mean()), float((sign(a["g_vis"]) != sign(b["g_vis"])).mean())]
                           for a, b in zip(tr32, tr64)])
    floor_loss = np.array([abs(a["loss"] - b["loss"]) / abs(b["loss"]) for a, b in zip(tr32, tr64)])
    floor_delta = np.array([float((np.abs(npy(r32["d_ir"]) - npy(od_ir)) > 1e-6).mean()),
                            float((np.abs(npy(r32["d_vis"]) - npy(od_vis)) > 1e-6).mean())])
    print("reference fp32 vs fp64, PGD-10 2x64x96 mit_b0:")
    print("  loss rel. error per iteration:", np.array2string(floor_loss, precision=2))
    print("  sign-mismatch fraction (ir, vis) per iteration:\n", np.array2string(floor_sign, precision=5))
    print("  differing delta elements (ir, vis):", floor_delta, " moved pixels:", int((npy(r32["pred"]) != npy(pred64)).sum()),
          " mIoU %.5f vs %.5f" % (float(np.nanmean(iou)), float(np.nanmean(iou64))))
    share = lambda p: (np.bincount(npy(p).ravel(), minlength=9) / float(npy(p).size)).astype(np.float32)
    print("  clean prediction: class shares", share(pred_clean), "mIoU %.5f;  attacked: class shares" % float(np.nanmean(O.compute_results(conf_clean)[2])),
          share(r32["pred"]), " pixels the attack moved:", int((npy(pred_clean) != npy(r32["pred"])).sum()), "of", npy(pred_clean).size)
    save("gn_attack_PGD10", d0_ir=npy(d0_ir), d0_vis=npy(d0_vis), delta_ir=npy(r32["d_ir"]), delta_vis=npy(r32["d_vis"]),
         gsum_ir=npy(r32["g_ir"]), gsum_vis=npy(r32["g_vis"]), losses=r32["losses"],
         sign_ir_per_iter=np.stack([sign(s["g_ir"]) for s in tr32]), sign_vis_per_iter=np.stack([sign(s["g_vis"]) for s in tr32]),
         sign64_ir_per_iter=np.stack([sign(s["g_ir"]) for s in tr64]), sign64_vis_per_iter=np.stack([sign(s["g_vis"]) for s in tr64]),
         losses64=np.array([s["loss"] for s in tr64]), delta64_ir=npy(od_ir).astype(np.float32), delta64_vis=npy(od_vis).astype(np.float32),
         pred=npy(r32["pred"]).astype(np.uint8), pred64=npy(pred64).astype(np.uint8), conf=conf, conf64=conf64,
         pred_clean=npy(pred_clean).astype(np.uint8), conf_clean=conf_clean, class_share=share(r32["pred"]), class_share_clean=share(pred_clean),
         logits=npy(r32["logits"]), fused=npy(r32["fused"]),
         floor_sign=floor_sign, floor_loss=floor_loss, floor_delta=floor_delta)


if __name__ == "__main__":
    main()
