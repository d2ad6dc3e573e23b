"""Fusion loss used by the (absent) training loop -- counterpart of the reference's core/loss.py:490-502
(`Fusionloss_grad2` = L1(mask, fused) + 1.1 * (1 - SSIM_11x11(fused, mask))).  Both entry scripts import and
instantiate it (test_original.py:10, robust_test.py:261) but never call it: evaluation does not need it.
The class is constructible for that reason; its forward belongs to BASELINE config 5 (training step) whose
kernels (SSIM, weight gradients, AdamW) are not built yet, so calling it raises."""
import torch.nn as nn


class Fusionloss_grad2(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, image_ir, image_vis, generate_img, mask):
        raise NotImplementedError("Fusionloss_grad2.forward: training-step kernels (SSIM loss, config 5) are not built yet")
