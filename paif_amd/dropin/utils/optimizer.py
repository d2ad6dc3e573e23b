"""`from utils.optimizer import PolyWarmupAdamW` (reference test_original.py:26)."""
from paif_amd.utils.optimizer import PolyWarmupAdamW, PolyWarmupAdamW_seg  # noqa: F401
