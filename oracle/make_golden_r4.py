"""TEST INFRASTRUCTURE -- golden vectors added in round 4.  Run ONLY in the build container:   python oracle/make_golden_r4.py

Same rules as oracle/make_golden.py: imports the real reference from /root/reference (oracle/ref_import.py + oracle/shims),
formula weights and inputs (paif_amd/synthetic.py, calibrated segmentation head), stores the REFERENCE's outputs as data.

go_forward_object_2x64x96: `Network_MM_Searched.forward_object` and `_detection_loss` (core/model_fusion_auto.py:1067-1097,
:1123-1128; the same code as Network_MM_CompModel's :736-766, :796-800) on 2 pairs of 64x96 through mit_b0, both composite
classes: the min-max-normalised fused plane, the logits, the loss value, its input gradients and a sample of its parameter
gradients (eval mode: running statistics, no dropout).
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
from oracle.make_golden import t, npy, save  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402


def main():
    R = ref_import.load()
    mfa = R["mfa"]
    torch.set_num_threads(8)
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    with ref_import.quiet():
        m = mfa.Network_MM_Searched(32, O.FUSION_AT, None, ce, "mit_b0", num_classes=9)
        fus = mfa.Network_Fusion_Searched(32, None, O.FUSION_AT)
        mc = mfa.Network_MM_CompModel(fus, None, ce, "mit_b0", 9, 256, None)
    head = S.head_tag("mit_b0", 2, 64, 96)
    for mod in (m, mc):
        mod.eval()
        S.load_formula_weights(mod, head=head)
    ir, vis, lab = S.make_batch(2, 64, 96)
    with torch.no_grad():
        fused, seg = m.forward_object(t(ir), t(vis))
        fused_c, seg_c = mc.forward_object(t(ir), t(vis))
    assert torch.equal(fused, fused_c) and torch.equal(seg, seg_c)          # the two classes run the same code on the same weights
    irg, visg = t(ir).requires_grad_(True), t(vis).requires_grad_(True)
    loss = m._detection_loss(irg, visg, t(lab))
    loss.backward()
    # the oracle's restatement against the reference, before anything is stored
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        of, os_ = O.model_forward_object(t(ir), t(vis), sd, "mit_b0")
    assert float((of - fused).abs().max()) <= 1e-5 and float((os_ - seg).abs().max()) <= 1e-5
    grads = {}
    for k, p in m.named_parameters():
        if p.grad is None:
            grads[k + "#none"] = np.zeros(0, np.float32)
            continue
        a = npy(p.grad).reshape(-1).astype(np.float32)
        grads[k + "#s"] = a[S.sample_indices(a.size)]
        grads[k + "#n"] = np.array(np.sqrt((a.astype(np.float64) ** 2).sum()))
    print("forward_object: fused plane in [%.3f, %.3f], share of clamped pixels %.3f, loss %.6f" %
          (float(fused.min()), float(fused.max()), float((fused == 0).float().mean()), float(loss)))
    save("go_forward_object_2x64x96", fused=npy(fused), logits=npy(seg), loss=np.array(float(loss)), d_ir=npy(irg.grad), d_vis=npy(visg.grad),
         **grads)


if __name__ == "__main__":
    main()
