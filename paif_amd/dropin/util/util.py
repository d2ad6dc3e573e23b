"""`from util.util import compute_results, visualize` (reference test_original.py:20, robust_test.py:17)."""
from paif_amd.util.util import ConfusionMeter, compute_results  # noqa: F401


def visualize(*args, **kwargs):
    """util/util.py:22-29 writes palette PNGs of the predictions: file I/O, out of scope (`visualize`).
    Both entry scripts import it; only commented-out lines call it."""
    raise NotImplementedError("util.util.visualize (palette PNG writer) is out of scope for paif_amd (SURVEY.md section 2); "
                              "use paif_amd.harness.write_fused_pngs for the fused images")
