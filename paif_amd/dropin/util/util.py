"""`from util.util import compute_results` (reference test_original.py:24)."""
from paif_amd.util.util import ConfusionMeter, compute_results  # noqa: F401
