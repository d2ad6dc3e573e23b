"""CPU: the oracle's TRAINING path (parameter gradients of _loss_coupled, train-mode BatchNorm / DropPath / Dropout2d,
PolyWarmupAdamW) against golden vectors produced by the reference's own modules + torch autograd
(oracle/make_golden_r2.py -> tests/golden/gm_*.npz).  This pins the checker the GPU tests of the training step use."""
import numpy as np
import pytest
import torch

from oracle import paif_oracle as O
from paif_amd import synthetic as S
from tests import helpers as Hh
from tests.helpers import t

EPS = 8 / 255.0
LR, WD, BETAS = 8e-5, 0.01, (0.9, 0.999)
SCHED = dict(warmup_iter=3000, max_iter=160000, warmup_ratio=1e-5, power=1.0)
START_STEP, DROP_SEED = 4000, 1234


def training_inputs():
    ir, vis, lab = S.make_batch(2, 64, 96)
    ir_adv = np.clip(ir + S.make_delta0(7, ir.shape, EPS), 0, 1).astype(np.float32)
    vis_adv = np.clip(vis + S.make_delta0(107, vis.shape, EPS), 0, 1).astype(np.float32)
    mask = np.maximum(ir, vis[:, :1]).astype(np.float32)
    return ir, vis, lab, ir_adv, vis_adv, mask


def group_of(name):
    """Parameter group index under the recipe of make_golden_r2.param_groups (WeTr.get_param_groups + the fusion net)."""
    if name.startswith("enhance_net."):
        return 3
    if name.startswith("denoise_net.encoder."):
        return 1 if "norm" in name[len("denoise_net.encoder."):] else 0
    return 2        # decoder + classifier


GROUP_LR_WD = [(LR, WD), (LR, 0.0), (LR * 10, WD), (LR, WD)]
NEVER = ("denoise_net.classifier.weight", "enhance_net.decompation.relu.weight")
# Some parameters have an EXACTLY zero gradient: a bias in front of a train-mode BatchNorm (the batch mean removes it:
# decoder.linear_c*.proj.bias, encoder.norm4.bias) and the key half of attn.kv.bias (softmax is invariant to a per-query
# constant).  Both sides hold only rounding noise there (|g| ~ 1e-8) and Adam turns noise into +-lr steps, so the movement
# comparison looks only at elements whose reference gradient is not noise.
ATOL = 1e-7


def movement_mismatch(golden, name, moved, base_flat, lr):
    """(#sampled elements whose 2-step movement differs from the reference's by > 5 % of lr, #elements compared)."""
    idx = S.sample_indices(base_flat.size)
    g0 = golden["grad0/" + name + "#s"]
    live = np.abs(g0) > max(1e-3 * np.abs(g0).max(), 1e-6)
    ref = golden["param2/" + name + "#s"] - base_flat[idx]
    return int((np.abs(moved[idx] - ref)[live] > 0.05 * lr).sum()), int(live.sum())


def trainable_sd(dtype=torch.float32):
    sd = Hh.model_sd("mit_b0")
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    for k in names:
        sd[k] = sd[k].to(dtype).clone().requires_grad_(True)
    for k in sd:
        if "running_" in k:
            sd[k] = sd[k].to(dtype).clone()
    return sd, names


def check_sampled(golden, prefix, named, rtol, what):
    for k, v in named:
        if prefix + k + "#none" in golden:
            assert v is None, "%s: %s should have no gradient" % (what, k)
            continue
        ref = golden[prefix + k + "#s"]
        assert v is not None, "%s: %s missing" % (what, k)
        a = v.detach().reshape(-1).numpy().astype(np.float32)[S.sample_indices(v.numel())]
        scale = max(float(golden[prefix + k + "#n"]) / np.sqrt(max(v.numel(), 1)), 1e-12)      # rms of the reference tensor
        assert np.abs(a - ref).max() <= rtol * max(scale, np.abs(ref).max()) + ATOL, (what, k, np.abs(a - ref).max(), scale)
        n = float(torch.sqrt((v.detach().double() ** 2).sum()))
        assert abs(n - float(golden[prefix + k + "#n"])) <= 10 * rtol * float(golden[prefix + k + "#n"]) + ATOL * np.sqrt(v.numel()), (what, k)


def test_oracle_param_grads_eval_mode(golden):
    g = golden("gm_param_grads_eval_mit_b0_2x64x96")
    ir, vis, lab, ir_adv, vis_adv, mask = training_inputs()
    sd, names = trainable_sd()
    loss = O.loss_coupled(t(ir_adv), t(vis_adv), t(mask), t(lab), sd, "mit_b0")
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    check_sampled(g, "", ((k, sd[k].grad) for k in names), 2e-4, "eval-mode grads")
    for k in NEVER:
        assert sd[k].grad is None and k + "#none" in g


def test_oracle_train_mode_two_optimizer_steps(golden):
    g = golden("gm_train_step_mit_b0_2x64x96")
    ir, vis, lab, ir_adv, vis_adv, mask = training_inputs()
    sd, names = trainable_sd()
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    try:
        for step in range(2):
            O.TRAIN = O.TrainCtx(DROP_SEED, rank=0, step=step)
            for k in names:
                sd[k].grad = None
            loss = O.loss_coupled(t(ir_adv), t(vis_adv), t(mask), t(lab), sd, "mit_b0")
            loss.backward()
            assert abs(float(loss) - float(g["losses"][step])) <= 2e-5 * abs(float(g["losses"][step])), (step, float(loss))
            if step == 0:
                check_sampled(g, "grad0/", ((k, sd[k].grad) for k in names), 2e-3, "train-mode grads")   # fp32 cancellation in pooled / batch-statistic gradients
                for k in sd:
                    if "running_" in k:
                        assert np.abs(sd[k].numpy() - g["bn0/" + k + "#b"]).max() <= 1e-6, k
            mult = O.poly_warmup_lr_mult(START_STEP + step, **SCHED)
            with torch.no_grad():
                for gi, (lr0, wd) in enumerate(GROUP_LR_WD):
                    ks = [k for k in names if group_of(k) == gi]
                    assert abs(lr0 * mult - float(g["lrs"][step][gi])) <= 1e-12
                    O.adamw_step([sd[k] for k in ks], [sd[k].grad for k in ks], [m[k] for k in ks], [v[k] for k in ks], step + 1,
                                 lr0 * mult, wd, BETAS)
    finally:
        O.TRAIN = None
    # parameters after two steps: Adam's first steps move every weight by ~lr * sign(g): compare the MOVEMENT
    sd0 = Hh.model_sd("mit_b0")
    bad = tot = 0
    for k in names:
        if k in NEVER:
            assert np.array_equal(sd[k].detach().numpy(), sd0[k].numpy())       # untouched: no gradient, no weight decay
            continue
        b, n = movement_mismatch(g, k, (sd[k].detach() - sd0[k]).reshape(-1).numpy(), sd0[k].reshape(-1).numpy(), GROUP_LR_WD[group_of(k)][0])
        bad, tot = bad + b, tot + n
    assert tot > 40000 and bad <= 0.005 * tot, (bad, tot)
    for k in sd:
        if "running_" in k:
            # after a step the zero-gradient biases in front of the head's BatchNorm have moved by +-lr (noise sign, see above):
            # the batch mean, hence running_mean, absorbs that shift -- 0.1 * 10 * lr * |W| ~ 1e-4
            assert np.abs(sd[k].numpy() - g["bn2/" + k + "#b"]).max() <= 1e-3, k
