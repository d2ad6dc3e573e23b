"""`from core import Fusionloss_grad2` (reference core/__init__.py re-exports loss.*, test_original.py:10)."""
from paif_amd.core.loss import Fusionloss_grad2  # noqa: F401
from paif_amd.core.mix_transformer import *  # noqa: F401,F403
from paif_amd.core.segformer_head import SegFormerHead  # noqa: F401
from paif_amd.core.model_fusion_auto import WeTr  # noqa: F401
