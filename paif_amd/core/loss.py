"""Fusion loss of the training API -- counterpart of the reference's core/loss.py:490-502 and pytorch_ssim/__init__.py:
`Fusionloss_grad2` = L1(mask, fused) + 1.1 * (1 - SSIM_11x11(fused, mask)).  Both entry scripts import and instantiate it
(test_original.py:10, robust_test.py:261) but never call it.

Built: the forward value and the gradient w.r.t. the generated image (HIP kernels paif_ssim_l1_fwd / _bwd_input), so the loss
can sit on top of the models' input-gradient autograd nodes.  Not built: parameter gradients -- the training step (BASELINE
config 5: wgrad kernels, AdamW kernels, RCCL gradient all-reduce) is outside round 1.  A `mask` that requires grad raises."""
import torch
import torch.nn as nn

from .. import ops


class _SsimL1Fn(torch.autograd.Function):
    """w_l1 * mean|y - x| + w_ss * (1 - mean SSIM(x, y)), differentiable w.r.t. x only."""

    @staticmethod
    def forward(ctx, x, y, w_l1, w_ss):
        xd, yd = x.detach(), y.detach()
        s, l1 = ops.ssim_l1(xd, yd)
        ctx.save_for_backward(xd, yd)
        ctx.w = (w_l1, w_ss)
        return w_l1 * l1 + w_ss * (1 - s)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        n = float(x.numel())
        dx = ops.ssim_l1_bwd(x, y, g * (ctx.w[0] / n), g * (ctx.w[1] / n))
        return dx, None, None, None


def _ssim_l1(x, y, w_l1, w_ss):
    if torch.is_grad_enabled() and y.requires_grad:
        raise NotImplementedError("ssim / Fusionloss_grad2: the gradient w.r.t. the second image (mask) is not built")
    if torch.is_grad_enabled() and x.requires_grad:
        return _SsimL1Fn.apply(x, y, w_l1, w_ss)
    with torch.no_grad():
        s, l1 = ops.ssim_l1(x, y)
        return w_l1 * l1 + w_ss * (1 - s)


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim.ssim for single-channel images (the only use on the path), mean over the batch."""
    if window_size != 11 or not size_average or img1.shape[1] != 1:
        raise NotImplementedError("ssim: built for window_size=11, size_average=True, 1 channel (core/loss.py:501)")
    return 1 - _ssim_l1(img1, img2, 0.0, 1.0)


class Fusionloss_grad2(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, image_ir, image_vis, generate_img, mask):
        return _ssim_l1(generate_img, mask[:, :1, :, :], 1.0, 1.1)
