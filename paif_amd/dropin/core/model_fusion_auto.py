"""`from core.model_fusion_auto import Network_Fusion_Searched, ...` (reference test_original.py:14-17,703)."""
from paif_amd.core.model_fusion_auto import *  # noqa: F401,F403
from paif_amd.core.model_fusion_auto import (Cell_Chain, Cell_Decom, MixedOp, Network_Fusion_Searched,  # noqa: F401
                                             RGB2YCrCb, YCrCb2RGB, spatial_attn_layer_M)
