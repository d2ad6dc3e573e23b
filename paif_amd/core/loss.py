"""Fusion loss of the training API -- counterpart of the reference's core/loss.py:490-502 and pytorch_ssim/__init__.py:
`Fusionloss_grad2` = L1(mask, fused) + 1.1 * (1 - SSIM_11x11(fused, mask)).  Both entry scripts import and instantiate it
(test_original.py:10, robust_test.py:261) but never call it.

Built: the forward VALUE (one HIP kernel, paif_ssim_l1_fwd).  Not built: its gradient -- the training step (BASELINE
config 5: parameter gradients, AdamW kernels, RCCL gradient all-reduce) is outside round 1, so an input that requires
grad raises instead of returning a tensor that silently cannot back-propagate."""
import torch
import torch.nn as nn

from .. import ops


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim.ssim for single-channel images (the only use on the path), mean over the batch."""
    if window_size != 11 or not size_average or img1.shape[1] != 1:
        raise NotImplementedError("ssim: built for window_size=11, size_average=True, 1 channel (core/loss.py:501)")
    ops.require_no_grad(img1, img2)
    return ops.ssim_l1(img1, img2)[0]


class Fusionloss_grad2(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, image_ir, image_vis, generate_img, mask):
        ops.require_no_grad(generate_img, mask)
        with torch.no_grad():
            s, l1 = ops.ssim_l1(generate_img, mask[:, :1, :, :])
            return l1 + 1.1 * (1 - s)
