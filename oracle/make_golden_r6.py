"""TEST INFRASTRUCTURE -- golden vectors added in round 6.  Run ONLY in the build container:   python oracle/make_golden_r6.py

Same rules as oracle/make_golden.py: imports the real reference from /root/reference (oracle/ref_import.py + oracle/shims),
formula weights and inputs (paif_amd/synthetic.py, the calibrated 480x640 head), stores the REFERENCE's outputs as data.

gq_model_b3_32x480x640: the reference's `Network_MM_Searched` (mit_b3) on THIRTY-TWO synthetic 480x640 pairs (`S.make_batch(32, 480,
640)`: samples 0..7 are the benchmarked batch of gp_model_b3_8x480x640, 8..31 continue the same index-seeded formula), one forward per
pair (B = 1 like the reference harness, test_original.py:111): the x4-upsampled argmax map in float32 and in float64 (the reference's
own noise floor), the confusion matrix against the synthetic labels, logit range and top-2 margin statistics.
Why: SURVEY 8(d)'s "argmax agreement >= 99.9 %" measured 99.929 % over eight samples with two of them below the line
taken alone -- eight samples cannot settle a 0.03-point margin.  tests/test_f16_storage_gpu.py evaluates the clause on these 9.83 M
pixels and asserts the LOWER end of a bootstrap interval over samples.
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

N = 32


def main():
    R = ref_import.load()
    torch.set_num_threads(8)
    m = build_model(R, "mit_b3", head=S.head_tag("mit_b3", 1, 480, 640))
    preds, preds64, confs, rng, med, shares = [], [], [], [], [], []
    for i in range(N):
        ir, vis, lab = S.make_batch(1, 480, 640, start=i)
        with torch.no_grad():
            f, s = m(t(ir), t(vis))
            up = torch.nn.functional.interpolate(s, size=(480, 640), mode="bilinear", align_corners=False)
        pred = npy(up.argmax(1))[0]
        m.double()
        torch.set_default_dtype(torch.float64)      # YCrCb2RGB builds its matrix with torch.tensor(...) (core/model_fusion_auto.py:96-100)
        with torch.no_grad():
            f64, s64 = m(t(ir).double(), t(vis).double())
        torch.set_default_dtype(torch.float32)
        m.float()
        pred64 = npy(torch.nn.functional.interpolate(s64, size=(480, 640), mode="bilinear", align_corners=False).argmax(1))[0]
        srt = np.sort(npy(up)[0], axis=0)
        r = float(s.max() - s.min())
        preds.append(pred.astype(np.uint8)); preds64.append(pred64.astype(np.uint8))
        confs.append(confusion_matrix(lab, pred[None]))
        rng.append(r); med.append(float(np.median(srt[-1] - srt[-2]) / r)); shares.append(class_share(pred))
        print("sample %d: classes >= 4 %%: %d, median top-2 margin %.2f %% of the logit range %.3f, float32 vs float64 pixels %d, mIoU %.4f" % (
            i, int((shares[-1] >= 0.04).sum()), 100 * med[-1], r, int((pred != pred64).sum()),
            float(np.nanmean(R["util"].compute_results(confs[-1])[2]))), flush=True)
    save("gq_model_b3_32x480x640", pred=np.stack(preds), pred64=np.stack(preds64), conf=np.stack(confs), logit_range=np.array(rng, np.float32),
         median_margin_over_range=np.array(med, np.float32), class_share=np.stack(shares).astype(np.float32))


if __name__ == "__main__":
    main()
