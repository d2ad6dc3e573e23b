"""`import operations_m` / `from operations_m import *` (reference test_original.py:6, core/model_fusion_auto.py:393)."""
from paif_amd.operations_m import *  # noqa: F401,F403
from paif_amd.operations_m import OPS, BasicConv, conv3x3  # noqa: F401
