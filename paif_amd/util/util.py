"""Metrics of the evaluation harness -- counterpart of the reference's util/util.py:31-55 and of the
sklearn.metrics.confusion_matrix(labels=0..8) call at test_original.py:209-211."""
import numpy as np
import torch

from .. import ops


def compute_results(conf_total):
    """util/util.py:31-55: per-class precision, recall, IoU from a confusion matrix (rows = label, cols = prediction);
    NaN for empty classes.  Integer counts -> float64 ratios on the host (9x9: negligible)."""
    conf_total = np.asarray(conf_total)
    n_class = conf_total.shape[0]
    precision_per_class = np.zeros(n_class)
    recall_per_class = np.zeros(n_class)
    iou_per_class = np.zeros(n_class)
    for cid in range(n_class):
        col, row, tp = conf_total[:, cid].sum(), conf_total[cid, :].sum(), conf_total[cid, cid]
        precision_per_class[cid] = np.nan if col == 0 else float(tp) / float(col)
        recall_per_class[cid] = np.nan if row == 0 else float(tp) / float(row)
        iou_per_class[cid] = np.nan if (row + col - tp) == 0 else float(tp) / float(row + col - tp)
    return precision_per_class, recall_per_class, iou_per_class


class ConfusionMeter:
    """Accumulates the confusion matrix ON THE GPU (integer atomics: exact, order independent) from segmentation
    logits and labels: bilinear x4 upsample + argmax (test_original.py:180,207) then counts (:209-211)."""

    def __init__(self, n_class=9, device="cuda"):
        self.n_class = n_class
        self.conf = torch.zeros((n_class, n_class), dtype=torch.int64, device=device)

    def update(self, seg_map, label):
        """seg_map [B,ncls,h,w] (NCHW view or contiguous), label int64 [B,H,W].  Returns the argmax map [B,H,W]."""
        logits = ops.to_nhwc(seg_map)
        pred = ops.upsample_argmax(logits, label.shape[1], label.shape[2])
        ops.confusion_matrix_accum_(self.conf, label.to(torch.int64), pred, self.n_class)
        return pred

    def results(self):
        return compute_results(self.conf.cpu().numpy())
