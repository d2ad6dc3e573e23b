"""`from attack.attack import attack_both, ...` (reference test_original.py:23, robust_test.py:23)."""
from paif_amd.attack.attack import (Seg_loss, attack_both, attack_ir, attack_vis, clamp, cos_pgd, fgsm_ir,  # noqa: F401
                                    pgd_attack_ir, pgd_attack_vision, seg_pgd, trans_format)
