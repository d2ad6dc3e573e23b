"""`from utils import eval_seg` (reference test_original.py:24, robust_test.py:21): imported by both entry scripts, never called
(out of scope).  The metric the harness uses is util.util.compute_results."""


def scores(*args, **kwargs):
    raise NotImplementedError("utils.eval_seg.scores is imported but never called by the reference's entry scripts; out of scope "
                              "(SURVEY.md section 2) -- use util.util.compute_results / paif_amd.util.util.ConfusionMeter")


_fast_hist = scores
