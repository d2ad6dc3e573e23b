"""TEST INFRASTRUCTURE -- golden vectors added in round 2 (training step, remaining attack branches).  Run ONLY in the
build container:   python oracle/make_golden_r2.py

Same rules as oracle/make_golden.py: imports the real reference from /root/reference (oracle/ref_import.py + oracle/shims),
formula weights and inputs (paif_amd/synthetic.py), stores the REFERENCE's outputs as small data fixtures.

The reference ships no training loop: the step pinned here is the one its API implies --
  loss = model._loss_coupled([ir_adv, ir], [vis_adv, vis], mask, labels)   (core/model_fusion_auto.py:1102-1109)
  loss.backward(); PolyWarmupAdamW.step()                                   (utils/optimizer.py:3-33)
with Fusionloss_grad2 / CrossEntropyLoss(ignore_index=255), run by the reference's own modules and torch autograd.
Stochastic layers (timm DropPath -- a shim, the package is not installable -- and nn.Dropout2d) are given the
counter-based keep masks of oracle.paif_oracle.TrainCtx, the stream the product uses, so the run is reproducible.
Big tensors are stored as an evenly spaced sample (synthetic.sample_indices) plus their L2 norm.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import paif_oracle as O  # noqa: E402
from oracle.make_golden import t, npy, save  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402

EPS, ALPHA = 8 / 255.0, 2 / 255.0
LR, WD, BETAS = 8e-5, 0.01, (0.9, 0.999)           # configs/voc.yaml:12-31
SCHED = dict(warmup_iter=3000, max_iter=160000, warmup_ratio=1e-5, power=1.0)
START_STEP = 4000                                    # past the warm-up, so the update is visible in fp32
DROP_SEED = 1234


def training_inputs():
    ir, vis, lab = S.make_batch(2, 64, 96)
    ir_adv = np.clip(ir + S.make_delta0(7, ir.shape, EPS), 0, 1).astype(np.float32)
    vis_adv = np.clip(vis + S.make_delta0(107, vis.shape, EPS), 0, 1).astype(np.float32)
    mask = np.maximum(ir, vis[:, :1]).astype(np.float32)
    return ir, vis, lab, ir_adv, vis_adv, mask


def param_groups(model):
    """The SegFormer recipe WeTr.get_param_groups() is written for (core/model_fusion_auto.py:44-60): encoder weights,
    encoder norms without weight decay, decoder at 10x the learning rate; plus the fusion network."""
    g = model.denoise_net.get_param_groups()
    return [dict(params=g[0], lr=LR, weight_decay=WD), dict(params=g[1], lr=LR, weight_decay=0.0),
            dict(params=g[2], lr=LR * 10, weight_decay=WD), dict(params=list(model.enhance_net.parameters()), lr=LR, weight_decay=WD)]


def sampled(named):
    out = {}
    for k, v in named:
        if v is None:
            out[k + "#none"] = np.zeros(0, np.float32)
            continue
        a = npy(v).reshape(-1).astype(np.float32)
        out[k + "#s"] = a[S.sample_indices(a.size)]
        out[k + "#n"] = np.array(float(np.sqrt((a.astype(np.float64) ** 2).sum())))
    return out


def patch_stochastic_layers(R, model, ctx):
    """DropPath (shim class of the un-vendored timm) and the head's Dropout2d draw from `ctx` (TrainCtx) instead of torch's RNG."""
    import timm.models.layers as tl

    def drop_path_forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        return x * ctx[0].keep(x.shape[0], self.drop_prob).view((x.shape[0],) + (1,) * (x.ndim - 1))

    tl.DropPath.forward = drop_path_forward
    R["mit"].DropPath.forward = drop_path_forward
    drop = model.denoise_net.decoder.dropout

    def dropout2d_forward(x):
        if not drop.training:
            return x
        return x * ctx[0].keep(x.shape[0] * x.shape[1], drop.p).view(x.shape[0], x.shape[1], 1, 1)

    drop.forward = dropout2d_forward


def main():
    R = ref_import.load()
    mfa = R["mfa"]
    torch.manual_seed(0)

    # ---- G-g3: pgd_attack_ir with the SSIM loss (attack/attack.py:136-137,160-162); attack_vis / attack_ir the way
    #            robust_test.py:169-176 calls them (keywords, X_fusion=, attack_way='newPGD') --------------------------------
    with ref_import.quiet():
        m0 = mfa.Network_MM_Searched(32, O.FUSION_AT, None, None, "mit_b0", num_classes=9)
    m0.eval()
    S.load_formula_weights(m0)
    ir, vis, lab = S.make_batch(2, 64, 96)
    with torch.no_grad():
        fused = m0(t(ir), t(vis))[0]
        X_rgb = R["attack"].trans_format(fused, t(vis))
    out = {}
    cases = [
        ("pgd_attack_ir.l_ssim", lambda: R["attack"].pgd_attack_ir(m0, t(vis), t(ir), X_rgb, t(lab), EPS, ALPHA, 2, 1, "l_ssim"), ir.shape),
        ("attack_vis.newPGD", lambda: R["attack"].attack_vis(m0, X_vis=t(vis), X_ir=t(ir), X_fusion=fused, label=t(lab), attack_loss="l_seg",
                                                             attack_iters=2, epsilon=EPS, alpha=ALPHA, attack_mode="vis", attack_way="newPGD"), vis.shape),
        ("attack_ir.newPGD", lambda: R["attack"].attack_ir(m0, X_vis=t(vis), X_ir=t(ir), X_fusion=fused, label=t(lab), attack_loss="l_seg",
                                                           attack_iters=2, epsilon=EPS, alpha=ALPHA, attack_mode="ir", attack_way="newPGD"), ir.shape),
    ]
    for name, fn, shp in cases:
        torch.manual_seed(4321)
        d0 = torch.zeros(shp).uniform_(-EPS, EPS)
        torch.manual_seed(4321)
        for p_ in m0.parameters():
            p_.grad = None
        with ref_import.quiet(), torch.no_grad():
            d = fn()
        out[name + ".d0"] = npy(d0)
        out[name + ".delta"] = npy(d)
    save("gg3_attacks_round2", X_rgb=npy(X_rgb), **out)

    # ---- G-m: the training step (mit_b0, 2 x 64x96) ----------------------------------------------------------------------
    ir, vis, lab, ir_adv, vis_adv, mask = training_inputs()
    with ref_import.quiet():
        crit = R["loss"].Fusionloss_grad2()
        model = mfa.Network_MM_Searched(32, O.FUSION_AT, crit, torch.nn.CrossEntropyLoss(ignore_index=255), "mit_b0", num_classes=9)
    S.load_formula_weights(model)
    args = ((t(ir_adv), t(ir)), (t(vis_adv), t(vis)), t(mask), t(lab))

    # (a) eval mode: every parameter gradient of _loss_coupled
    model.eval()
    loss = model._loss_coupled(*args)
    loss.backward()
    save("gm_param_grads_eval_mit_b0_2x64x96", loss=np.array(float(loss)), **sampled((k, p.grad) for k, p in model.named_parameters()))

    # (b) train mode (BatchNorm batch statistics, DropPath, Dropout2d): gradients of step 0, then two optimizer steps
    for p_ in model.parameters():
        p_.grad = None
    S.load_formula_weights(model)
    model.train()
    ctx = [None]
    patch_stochastic_layers(R, model, ctx)
    opt = R["optimizer"].PolyWarmupAdamW(params=param_groups(model), lr=LR, weight_decay=WD, betas=list(BETAS), **SCHED)
    opt.global_step = START_STEP
    losses, lrs = [], []
    for step in range(2):
        ctx[0] = O.TrainCtx(DROP_SEED, rank=0, step=step)
        opt.zero_grad()
        loss = model._loss_coupled(*args)
        loss.backward()
        if step == 0:
            g0 = sampled((k, p.grad) for k, p in model.named_parameters())
            bn0 = {k + "#b": npy(v).copy() for k, v in model.state_dict().items() if "running_" in k}   # .numpy() aliases the buffer
        opt.step()
        losses.append(float(loss))
        lrs.append([g["lr"] for g in opt.param_groups])
    save("gm_train_step_mit_b0_2x64x96", losses=np.array(losses), lrs=np.array(lrs),
         **{"grad0/" + k: v for k, v in g0.items()}, **{"bn0/" + k: v for k, v in bn0.items()},
         **{"param2/" + k: v for k, v in sampled(model.named_parameters()).items()},
         **{"bn2/" + k + "#b": npy(v) for k, v in model.state_dict().items() if "running_" in k})
    print("done")


if __name__ == "__main__":
    main()
