"""PGD / segPGD / cosPGD adversarial attacks on both modalities -- MI355X-native counterpart of the
reference's attack/attack.py (attack_both :417-514, attack_vis :517-604, attack_ir :607-690, Seg_loss :103-114).

Same signatures and return values.  Semantics reproduced on purpose:
  * the reference never zeroes delta.grad, so the step uses sign(RUNNING SUM of gradients) (:501-512);
  * delta0 ~ U(-eps, eps) drawn from torch's global RNG, clamped so X + delta stays in [0,1] (:433-441);
  * an unknown attack_loss prints and returns -1 (:428-430).
Not reproduced: the reference also leaves .grad on all 44.9 M model parameters (wasted weight gradients);
here only input gradients are computed.

Every attack_way ('PGD' -- the only branch either entry script uses -- 'segPGD', 'cosPGD', 'newPGD') runs entirely on
hand-written HIP kernels: model forward with a tape, the variant's loss on the bilinearly upsampled logits and its gradient
(paif_attack_loss_fwd / _bwd: masked CE terms of segPGD, cosine factor of cosPGD incl. its own gradient through max_c),
hand-written reverse pass, fused PGD update.  A user model without `forward_taped` (not one of ours) goes through autograd
with the same HIP loss node.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

upper_limit = 1
lower_limit = 0


def clamp(X, lower_limit, upper_limit):
    """attack/attack.py:68-72."""
    return torch.max(torch.min(X, upper_limit), lower_limit)


class Seg_loss(nn.Module):
    """attack/attack.py:103-114: CrossEntropyLoss(ignore_index=255) on already-upsampled outputs [B,C,H,W].
    Runs the fused HIP kernels (paif_upsample_ce_fwd / _bwd; identity-size bilinear sampling is exact), with or
    without a gradient (ops.UpsampleCE is the autograd node)."""

    def __init__(self):
        super().__init__()
        self._loss = torch.nn.CrossEntropyLoss(ignore_index=255)   # kept for attribute parity; never evaluated

    def forward(self, outputs, labels):
        return ops.upsample_ce(outputs, labels.type(torch.long).contiguous(), ignore_index=255)


def _loss_variant(seg_map, label, attack_way, i, attack_iters):
    """attack/attack.py:445-499 on the LOW-resolution seg_map: F.interpolate(..., bilinear) + the attack_way branch, as one HIP
    autograd node (ops.AttackLoss)."""
    return ops.attack_loss(seg_map, label, attack_way, i, attack_iters)


def _init_delta(X, epsilon, delta0):
    if delta0 is None:
        d = torch.zeros_like(X).uniform_(-epsilon, epsilon)
    else:
        d = delta0.to(X.device, torch.float32).clone()
    # clamp(delta, 0 - X, 1 - X) (:435,440) -- the PGD-step kernel with alpha = 0 is exactly that clamp
    return ops.pgd_step_(d.contiguous(), d, X, 0.0, float("inf"))


def _check_labels(lab64, model):
    """torch's CrossEntropyLoss (the reference's Seg_loss, attack/attack.py:103-114) raises on a label outside [0, C) that is not
    ignore_index; the loss kernels never index with such a value (they drop the pixel), so the range is validated here: ONCE per
    attack call, before the loop.  One host read; impossible while the stream is being captured into a hipGraph -- a caller that
    captures an attack validates its labels beforehand (bench.py's eager warm-up step does; `validate_labels` is the public form)."""
    ncls = getattr(getattr(model, "denoise_net", None), "num_classes", None)
    if ncls is None or torch.cuda.is_current_stream_capturing():
        return
    validate_labels(lab64, ncls)


def validate_labels(label, num_classes, ignore_index=255):
    lab = label.type(torch.long)
    bad = (lab != ignore_index) & ((lab < 0) | (lab >= num_classes))
    n = int(bad.sum())
    if n:
        raise RuntimeError("attack: %d label value(s) outside [0, %d) that are not ignore_index %d" % (n, num_classes, ignore_index))


def _attack(model, X_vis, X_ir, label, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, do_ir, do_vis,
            delta0_ir=None, delta0_vis=None, trace=None):
    if attack_loss == 'l_seg':
        criterion = Seg_loss()
    elif attack_loss == 'l_2':
        criterion = nn.MSELoss()
    else:
        print('dont give the correct loss function')
        return -1
    X_vis = X_vis.contiguous().float()
    X_ir = X_ir.contiguous().float()
    label = label.contiguous()
    fast = attack_loss == 'l_seg' and hasattr(model, "forward_taped")
    lab64 = label.type(torch.long).contiguous()
    if fast:
        _check_labels(lab64, model)
    with ops.attack_arithmetic():          # exact-fp32 kernels for the whole loop unless ops.set_attack_precision("fast")
        return _attack_loop(model, X_vis, X_ir, label, lab64, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, do_ir, do_vis,
                            delta0_ir, delta0_vis, trace, criterion, fast)


def _attack_loop(model, X_vis, X_ir, label, lab64, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, do_ir, do_vis,
                 delta0_ir, delta0_vis, trace, criterion, fast):
    for _ in range(restarts):
        d_ir = _init_delta(X_ir, epsilon, delta0_ir) if do_ir else torch.zeros_like(X_ir)
        d_vis = _init_delta(X_vis, epsilon, delta0_vis) if do_vis else torch.zeros_like(X_vis)
        g_ir = torch.zeros_like(X_ir)     # the never-zeroed delta.grad (:501): a running sum over iterations
        g_vis = torch.zeros_like(X_vis)
        # exact power of two on the (linear) reverse pass, undone in the accumulation below; from the label's VALID pixels
        gs = ops.attack_grad_scale(lab64) if fast else 1.0
        for i in range(attack_iters):
            if fast:
                with torch.no_grad():
                    with ops.attack_forward_arithmetic():
                        _, logits, tape = model.forward_taped(ops.add(X_ir, d_ir), ops.add(X_vis, d_vis))
                    way, wt, wf = ops.attack_loss_weights(attack_way, i, attack_iters)
                    coef = ops.attack_loss_fwd(logits, lab64, way, wt, wf)    # loss + the backward's scalars, on the device
                    d32 = ops.attack_loss_bwd(logits, lab64, coef, way, wt, wf, upstream=gs)
                    with ops.attack_backward_arithmetic():
                        gi, gv = model.backward_taped(d32, tape)
                    loss = coef[0]
            else:
                xi = ops.add(X_ir, d_ir).requires_grad_(True)
                xv = ops.add(X_vis, d_vis).requires_grad_(True)
                with torch.enable_grad(), ops.no_param_grads():
                    _, seg_map = model(xi, xv)
                    if attack_loss == 'l_seg':
                        loss = _loss_variant(seg_map, label, attack_way, i, attack_iters)
                    else:   # 'l_2' (nn.MSELoss on the upsampled map against the label, :424-425): a user criterion, torch ops
                        loss = criterion(F.interpolate(seg_map, size=label.shape[1:], mode='bilinear', align_corners=False), label)
                gi, gv = torch.autograd.grad(loss, [xi, xv])
            with torch.no_grad():
                if do_ir:
                    ops.axpy_(g_ir, gi.contiguous(), 1.0 / gs)
                    ops.pgd_step_(d_ir, g_ir, X_ir, alpha, epsilon)
                if do_vis:
                    ops.axpy_(g_vis, gv.contiguous(), 1.0 / gs)
                    ops.pgd_step_(d_vis, g_vis, X_vis, alpha, epsilon)
            if trace is not None:
                trace.append(dict(loss=float(loss.detach()), g_ir=g_ir.clone(), g_vis=g_vis.clone()))
    if fast:
        ops.check_attack_range(g_ir, g_vis)      # fp16 pairs: an operand outside fp16's exponent range shows up here, loudly
    # the reference returns Variables that carry the accumulated .grad
    d_ir.requires_grad_(True)
    d_vis.requires_grad_(True)
    d_ir.grad, d_vis.grad = g_ir, g_vis
    return d_ir, d_vis


def attack_both(model, X_vis, X_ir, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50,
                restarts=1, attack_loss='l_seg', attack_mode='vis', attack_way='PGD', *, delta0_ir=None, delta0_vis=None, trace=None):
    """attack/attack.py:417-514.  Extra keyword-only extensions: delta0_* (deterministic start), trace (per-iteration log)."""
    return _attack(model, X_vis, X_ir, label, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, True, True,
                   delta0_ir, delta0_vis, trace)


def attack_vis(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50,
               restarts=1, attack_loss='l_seg', attack_mode='vis', attack_way='PGD', *, delta0_vis=None):
    """attack/attack.py:517-604: perturb the visible image only; returns delta_vis.  X_fusion is accepted and unused,
    as in the reference (its body never reads it); call sites: robust_test.py:169-176."""
    r = _attack(model, X_vis, X_ir, label, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, False, True,
                None, delta0_vis)
    return r if r == -1 else r[1]


def attack_ir(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50,
              restarts=1, attack_loss='l_seg', attack_mode='vis', attack_way='PGD', *, delta0_ir=None):
    """attack/attack.py:607-690: perturb the infrared image only; returns delta_ir.  X_fusion: accepted, unused."""
    r = _attack(model, X_vis, X_ir, label, epsilon, alpha, attack_iters, restarts, attack_loss, attack_way, True, False,
                delta0_ir, None)
    return r if r == -1 else r[0]


# ---------------------------------------------------------------------------------------------
# Single-modality attacks that neither entry script calls (attack/attack.py:117-411).
# They take a FRESH gradient every iteration (torch.autograd.grad, no accumulation).  The model forward/backward is
# the HIP autograd node, and so is the loss glue on top (trans_format, image-space / masked / cosine losses: ops.TransFormat,
# ops.ImageLoss, ops.AttackLoss).
# fgsm_ir (:247-304) is not provided: it cannot run in the reference either (UnboundLocalError on `black_X` at :295
# without the mask, undefined `map_generate3` in get_ir_mask (:232-244) with it).
# ---------------------------------------------------------------------------------------------
def trans_format(image_fusion, images_vis):
    """attack/attack.py:75-100: recompose RGB from the fused Y + the visible Cr/Cb, clamp, global min-max -- a HIP autograd node
    (ops.TransFormat: the composite model's own fusion->seg glue kernels + a per-channel affine), differentiable w.r.t. image_fusion."""
    return ops.trans_format(image_fusion, images_vis)


def _fresh_grad_attack(model, X_vis, X_ir, X_fusion, label, epsilon, alpha, attack_iters, restarts, mode, loss_fn, delta0):
    X = (X_vis if mode == 'vis' else X_ir).contiguous().float()
    with ops.attack_arithmetic():
        return _fresh_grad_loop(model, X, X_vis, X_ir, epsilon, alpha, attack_iters, restarts, mode, loss_fn, delta0)


def _fresh_grad_loop(model, X, X_vis, X_ir, epsilon, alpha, attack_iters, restarts, mode, loss_fn, delta0):
    delta = None
    for _ in range(restarts):
        delta = _init_delta(X, epsilon, delta0)
        for i in range(attack_iters):
            xa = ops.add(X, delta).requires_grad_(True)
            with torch.enable_grad(), ops.no_param_grads():
                fused, seg_map = model(X_ir, xa) if mode == 'vis' else model(xa, X_vis)
                loss = loss_fn(fused, seg_map, i)
            g = torch.autograd.grad(loss, [xa])[0]
            with torch.no_grad():
                ops.pgd_step_(delta, g.contiguous(), X, alpha, epsilon)
    delta.requires_grad_(True)
    return delta


def _image_or_seg_loss(attack_loss, X_vis, X_fusion, label, sign=1.0):
    if attack_loss == 'l_seg':
        # Seg_loss on the upsampled map (:134-135): the fused HIP upsample + CE node takes the low-resolution map directly
        return lambda fused, seg, i: ops.upsample_ce(seg, label.type(torch.long), ignore_index=255)
    if attack_loss in ('l_2', 'l_1'):        # nn.MSELoss() / nn.L1Loss() on the recomposed image (:132-133): HIP loss + gradient kernels
        return lambda fused, seg, i: ops.image_loss(trans_format(fused, X_vis), X_fusion, attack_loss, sign)
    if attack_loss == 'l_ssim':
        # pytorch_ssim.SSIM()(robust_output, X_fusion) (attack/attack.py:136-137): every channel uses the same 11x11 window
        # and the result is the mean over everything, so [B,3,H,W] is evaluated as 3B single-channel images (HIP SSIM
        # forward / gradient kernels).  X_fusion must have the recomposed image's 3 channels, as in pytorch_ssim.
        from ..core.loss import ssim as _ssim

        def crit(a, b):
            if a.shape != b.shape:
                raise RuntimeError("l_ssim: X_fusion %s must match the recomposed image %s (pytorch_ssim convolves both with "
                                   "the same per-channel window)" % (tuple(b.shape), tuple(a.shape)))
            H, W = a.shape[-2:]
            return _ssim(a.reshape(-1, 1, H, W), b.reshape(-1, 1, H, W))
    else:
        raise NotImplementedError("attack_loss %r: l_seg / l_2 / l_1 / l_ssim are built (lpips needs the un-vendored LPIPS "
                                  "network weights, SURVEY.md section 2)" % attack_loss)
    return lambda fused, seg, i: sign * crit(trans_format(fused, X_vis), X_fusion)


def pgd_attack_ir(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50, restarts=1,
                  attack_loss='l_2', delta0=None):
    """attack/attack.py:117-172."""
    return _fresh_grad_attack(model, X_vis, X_ir, X_fusion, label, epsilon, alpha, attack_iters, restarts, 'ir',
                              _image_or_seg_loss(attack_loss, X_vis, X_fusion, label), delta0)


def pgd_attack_vision(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50,
                      restarts=1, attack_loss='l_seg', delta0=None):
    """attack/attack.py:175-229 (the image-space losses enter with a minus sign there, :218)."""
    return _fresh_grad_attack(model, X_vis, X_ir, X_fusion, label, epsilon, alpha, attack_iters, restarts, 'vis',
                              _image_or_seg_loss(attack_loss, X_vis, X_fusion, label, sign=-1.0), delta0)


def seg_pgd(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50, restarts=1,
            attack_loss='l_seg', attack_mode='vis', delta0=None):
    """attack/attack.py:307-365."""
    def loss_fn(fused, seg, i):
        return _loss_variant(seg, label, 'segPGD', i, attack_iters)

    return _fresh_grad_attack(model, X_vis, X_ir, X_fusion, label, epsilon, alpha, attack_iters, restarts,
                              'vis' if attack_mode == 'vis' else 'ir', loss_fn, delta0)


def cos_pgd(model, X_vis, X_ir, X_fusion, label, epsilon=8 / 255., alpha=2 / 255., attack_iters=50, restarts=1,
            attack_loss='l_seg', attack_mode='vis', delta0=None):
    """attack/attack.py:368-411."""
    def loss_fn(fused, seg, i):
        return _loss_variant(seg, label, 'cosPGD', i, attack_iters)

    return _fresh_grad_attack(model, X_vis, X_ir, X_fusion, label, epsilon, alpha, attack_iters, restarts,
                              'vis' if attack_mode == 'vis' else 'ir', loss_fn, delta0)


def fgsm_ir(*args, **kwargs):
    raise NotImplementedError("fgsm_ir cannot run in the reference either (attack/attack.py:295 UnboundLocalError `black_X`; "
                              "with_mask=True needs the undefined map_generate3, :232-244)")
