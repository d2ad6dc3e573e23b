"""TEST INFRASTRUCTURE -- golden vectors added in round 5.  Run ONLY in the build container:   python oracle/make_golden_r5.py

Same rules as oracle/make_golden.py: imports the real reference from /root/reference (oracle/ref_import.py + oracle/shims),
formula weights and inputs (paif_amd/synthetic.py, the calibrated 480x640 head), stores the REFERENCE's outputs as data.

gp_model_b3_8x480x640: the reference's `Network_MM_Searched` (mit_b3) on each of the EIGHT synthetic 480x640 pairs of the benchmarked
batch (`S.make_batch(8, 480, 640)`), one forward per pair (B = 1: the glue's min-max is batch-global, the reference harness runs
B = 1, test_original.py:111): the x4-upsampled argmax map, the confusion matrix against the synthetic labels, the logit range and the
top-2 margin statistics of every sample, in float32 -- and the argmax of the float64 run (the reference's own noise floor).
Why: "argmax agreement" on ONE 480x640 sample is a noisy statistic at the 1e-3 level (the pixels that move are near-ties and come in
spatial clusters): a 16-bit storage format whose expected agreement is 99.92 % measures 99.80 ... 99.97 % sample to sample
(tools/storage_sensitivity.py --phase 4).  SURVEY 8(d)'s clause is evaluated on the 2.46 M pixels of the eight samples.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.make_golden import build_model, class_share, npy, save, t  # noqa: E402
from oracle.paif_oracle import confusion_matrix  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402


def main():
    R = ref_import.load()
    torch.set_num_threads(8)
    m = build_model(R, "mit_b3", head=S.head_tag("mit_b3", 1, 480, 640))
    ir, vis, lab = S.make_batch(8, 480, 640)
    preds, preds64, confs, rng, med, shares, fmax = [], [], [], [], [], [], []
    for i in range(8):
        with torch.no_grad():
            f, s = m(t(ir[i:i + 1]), t(vis[i:i + 1]))
            up = torch.nn.functional.interpolate(s, size=(480, 640), mode="bilinear", align_corners=False)
        pred = npy(up.argmax(1))[0]
        m.double()
        torch.set_default_dtype(torch.float64)      # YCrCb2RGB builds its matrix with torch.tensor(...) (core/model_fusion_auto.py:96-100)
        with torch.no_grad():
            f64, s64 = m(t(ir[i:i + 1]).double(), t(vis[i:i + 1]).double())
        torch.set_default_dtype(torch.float32)
        m.float()
        pred64 = npy(torch.nn.functional.interpolate(s64, size=(480, 640), mode="bilinear", align_corners=False).argmax(1))[0]
        srt = np.sort(npy(up)[0], axis=0)
        r = float(s.max() - s.min())
        preds.append(pred.astype(np.uint8)); preds64.append(pred64.astype(np.uint8))
        confs.append(confusion_matrix(lab[i:i + 1], pred[None]))
        rng.append(r); med.append(float(np.median(srt[-1] - srt[-2]) / r)); shares.append(class_share(pred)); fmax.append(float(f.abs().max()))
        print("sample %d: classes >= 4 %%: %d, median top-2 margin %.2f %% of the logit range %.3f, float32 vs float64 pixels %d, mIoU %.4f" % (
            i, int((shares[-1] >= 0.04).sum()), 100 * med[-1], r, int((pred != pred64).sum()),
            float(np.nanmean(R["util"].compute_results(confs[-1])[2]))), flush=True)
    save("gp_model_b3_8x480x640", pred=np.stack(preds), pred64=np.stack(preds64), conf=np.stack(confs), logit_range=np.array(rng, np.float32),
         median_margin_over_range=np.array(med, np.float32), class_share=np.stack(shares).astype(np.float32), fused_absmax=np.array(fmax, np.float32))


if __name__ == "__main__":
    main()
