"""`from core.model_fusion_auto import Network_Fusion_Searched, ...` (reference test_original.py:14-17,703)."""
from paif_amd.core.model_fusion_auto import (Cell_Chain, Cell_Decom, ChannelPool, MixedOp, Network_Fusion_Searched,  # noqa: F401
                                             Network_MM_CompModel, Network_MM_Searched, Network_MM_SearchedFusion,
                                             RGB2YCrCb, WeTr, YCrCb2RGB, spatial_attn_layer_M)
