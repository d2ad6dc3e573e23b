"""TEST INFRASTRUCTURE -- golden-vector generator.  Run ONLY in the build container:

    python oracle/make_golden.py            # writes tests/golden/*.npz, *.json

Imports the real reference from /root/reference (via oracle/ref_import.py + oracle/shims),
loads formula weights (paif_amd/synthetic.py), feeds formula inputs and stores the reference's
outputs as small fixtures.  The fixtures hold DATA only (inputs are regenerated from the
formulas; expected outputs are stored); no reference source travels.  The reference has no
tests/fixtures of its own, so these outputs of the reference itself are
what pins the oracle and the HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import paif_oracle as O  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

PRIMITIVES = [  # the [probe] list of SURVEY.md 8(a) O7 + the shipped genotype
    "Denseblocks_3_1", "DilConv_3_2", "ECAattention_3", "Residualblocks_7_1",
    "SPAattention_3", "SepConv_3_1", "SepConv_5_1", "DilConv_5_1",
    "Denseblocks_5_2", "Denseblocks_7_1", "Residualblocks_3_2", "Residualblocks_5_2",
]


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def npy(x):
    return x.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def build_model(R, backbone, cls="Network_MM_Searched", head=None):
    """`head`: calibration tag of the segmentation head's last layer (paif_amd/synthetic.py, oracle/calibrate_head.py) --
    every golden whose argmax / confusion matrix / mIoU is compared carries one, so that the reference's map is multi-class."""
    with ref_import.quiet():
        if cls == "Network_MM_Searched":
            m = R["mfa"].Network_MM_Searched(32, O.FUSION_AT, None, None, backbone, num_classes=9)
        else:
            fus = R["mfa"].Network_Fusion_Searched(32, None, O.FUSION_AT)
            m = R["mfa"].Network_MM_CompModel(fus, None, None, backbone, 9, 256, None)
    m.eval()
    S.load_formula_weights(m, head=head)
    return m


def class_share(pred):
    return (np.bincount(np.asarray(pred).ravel().astype(np.int64), minlength=9) / float(np.asarray(pred).size)).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = ref_import.load()
    mfa, ops = R["mfa"], R["ops"]
    torch.manual_seed(0)

    # ---- G-a: every primitive, forward + input gradient under loss = sum(y * r) -------------
    x_np = S.make_smooth_feature(11, 1, 32, 24, 32)
    r_np = S.make_feature(12, (1, 32, 24, 32))
    out = {}
    for prim in PRIMITIVES:
        with ref_import.quiet():
            op = mfa.MixedOp(32, prim)
        op.eval()
        S.load_formula_weights(op, salt=PRIMITIVES.index(prim) + 1)
        x = t(x_np).requires_grad_(True)
        y = op(x)
        (y * t(r_np)).sum().backward()
        out[prim + ".y"] = npy(y)
        out[prim + ".dx"] = npy(x.grad)
    save("ga_primitives", **out)

    # ---- G-b: guided filter (r=4, both eps) + fp64 variant documenting the fp32 noise floor --
    from guided_filter_pytorch.guided_filter import GuidedFilter

    y_np = S.make_smooth_feature(21, 1, 32, 24, 32)
    guide = t(y_np).max(1, keepdim=True)[0] - t(y_np).min(1, keepdim=True)[0]
    out = {}
    for eps in (1e-3, 1e-4):
        out["lf_eps%g" % eps] = npy(GuidedFilter(4, eps)(guide, t(y_np)))
        out["lf64_eps%g" % eps] = npy(GuidedFilter(4, eps)(guide.double(), t(y_np).double())).astype(np.float32)
    save("gb_guided_filter", **out)

    # ---- G-c: fusion net with intermediates ------------------------------------------------
    with ref_import.quiet():
        fus = mfa.Network_Fusion_Searched(32, None, O.FUSION_AT)
    fus.eval()
    S.load_formula_weights(fus)
    inter = {}
    hooks = [
        fus.stem_1.register_forward_hook(lambda m, i, o: inter.__setitem__("fir", npy(o))),
        fus.stem_2.register_forward_hook(lambda m, i, o: inter.__setitem__("fvis", npy(o))),
        fus.decompation.conv1x1_lf.register_forward_hook(lambda m, i, o: inter.__setitem__("lf", npy(o))),
        fus.decompation.conv1x1_hf.register_forward_hook(lambda m, i, o: inter.__setitem__("hf", npy(o))),
        fus.decompation.chain.register_forward_hook(lambda m, i, o: inter.__setitem__("lf_re", npy(o))),
        fus.decompation.chain2.register_forward_hook(lambda m, i, o: inter.__setitem__("hf_re", npy(o))),
        fus.decompation.register_forward_hook(
            lambda m, i, o: inter.update(ir_feature=npy(o[0]), vis_feature=npy(o[1]))),
        fus.spa.register_forward_hook(lambda m, i, o: inter.__setitem__("scale", npy(o))),
        fus.chain.register_forward_hook(lambda m, i, o: inter.__setitem__("feature2", npy(o))),
    ]
    ir, vis, _ = S.make_batch(1, 48, 64)
    ycc = mfa.RGB2YCrCb(t(vis))
    with torch.no_grad():
        fused = fus(t(ir), ycc[:, 0:1])
    keep = {k: inter[k] for k in ("fir", "lf", "hf", "lf_re", "ir_feature", "vis_feature", "scale", "feature2")}
    save("gc_fusion_48x64", fused=npy(fused), **keep)
    for h in hooks:
        h.remove()
    ir, vis, _ = S.make_batch(2, 64, 96)
    ycc = mfa.RGB2YCrCb(t(vis))
    irt = t(ir).requires_grad_(True)
    yt = ycc[:, 0:1].clone().requires_grad_(True)
    fused = fus(irt, yt)
    rr = S.make_feature(31, tuple(fused.shape))
    (fused * t(rr)).sum().backward()
    save("gc_fusion_2x64x96", fused=npy(fused), d_ir=npy(irt.grad), d_y=npy(yt.grad))

    # ---- G-j: feature-visualisation path (Network_Fusion_Searched_showfeatures.forward2, :669-679) ----
    with ref_import.quiet():
        show = mfa.Network_Fusion_Searched_showfeatures(32, None, O.FUSION_AT)
    show.eval()
    S.load_formula_weights(show)
    ir, vis, _ = S.make_batch(1, 40, 56)
    ycc = mfa.RGB2YCrCb(t(vis))
    with torch.no_grad():
        outs = show.forward2(t(ir), ycc[:, 0:1])
    names = ("fused", "ir_feature", "vis_feature", "lf_ir", "hf_ir", "res_ir", "lf_vis", "hf_vis", "res_vis")
    save("gj_showfeatures_40x56", **{n: npy(o) for n, o in zip(names, outs)})

    # ---- G-k: fused-image writer post-processing (test_original.py:181-197), by EXECUTING the reference's own lines ----
    import textwrap
    src = open(os.path.join(ref_import.REF_ROOT, "test_original.py")).read().split("\n")
    beg = next(i for i, l in enumerate(src) if l.strip() == "images_vis_ycrcb = RGB2YCrCb(images_vis)")
    end = next(i for i in range(beg, len(src)) if src[i].strip() == "for k in range(len(name)):")
    block = textwrap.dedent("\n".join(src[beg:end]))
    ir, vis, _ = S.make_batch(2, 48, 64)
    with torch.no_grad():
        ycc = mfa.RGB2YCrCb(t(vis))
        fused_in = fus(t(ir), ycc[:, 0:1])
    ns = dict(RGB2YCrCb=mfa.RGB2YCrCb, YCrCb2RGB=mfa.YCrCb2RGB, torch=torch, np=np, images_vis=t(vis), image_fusion=fused_in)
    exec(block, ns)
    assert ns["fused_image"].dtype == np.uint8 and ns["fused_image"].shape == (2, 48, 64, 3)
    save("gk_fused_writer_2x48x64", fused=npy(fused_in), fused_image=ns["fused_image"])

    # ---- G-l: training-API loss values (_loss, _loss_coupled, _fusion_loss_lower, _fusion_loss; :1093-1122) ----
    with ref_import.quiet():
        crit = R["loss"].Fusionloss_grad2()
        ml = mfa.Network_MM_Searched(32, O.FUSION_AT, crit, torch.nn.CrossEntropyLoss(ignore_index=255), "mit_b0", num_classes=9)
    ml.eval()
    S.load_formula_weights(ml)
    ir, vis, lab = S.make_batch(2, 64, 96)
    ir2, vis2, _ = S.make_batch(2, 64, 96, start=2)
    mask = np.maximum(ir, vis[:, :1]).astype(np.float32)      # any [B,1,H,W] target image
    with torch.no_grad():
        out = dict(
            loss=float(ml._loss(t(ir), t(vis), t(mask), t(lab))),
            loss_coupled=float(ml._loss_coupled((t(ir), t(ir2)), (t(vis), t(vis2)), t(mask), t(lab))),
            fusion_loss_lower=float(ml._fusion_loss_lower(t(ir), t(vis), t(mask))),
            fusion_loss=float(ml._fusion_loss(t(ir), t(vis), t(mask))),
        )
    # input gradients of _loss (loss gradient -> fused and seg_map -> both networks -> ir, vis)
    irg, visg = t(ir).requires_grad_(True), t(vis).requires_grad_(True)
    ml._loss(irg, visg, t(mask), t(lab)).backward()
    irl, visl = t(ir).requires_grad_(True), t(vis).requires_grad_(True)
    ml._fusion_loss_lower(irl, visl, t(mask)).backward()
    save("gl_training_losses_2x64x96", **{k: np.array(v, dtype=np.float64) for k, v in out.items()},
         loss_d_ir=npy(irg.grad), loss_d_vis=npy(visg.grad), lower_d_ir=npy(irl.grad), lower_d_vis=npy(visl.grad))

    # ---- G-d: colour transforms + clamp / batch-global min-max / normalise ------------------
    m0 = build_model(R, "mit_b0", head=S.head_tag("mit_b0", 2, 64, 96))
    ir, vis, _ = S.make_batch(2, 64, 96)
    with torch.no_grad():
        ycc = mfa.RGB2YCrCb(t(vis))
        rgb = mfa.YCrCb2RGB(ycc)
    # seg_in is internal to forward(): capture it as the input of denoise_net
    cap = {}
    h = m0.denoise_net.register_forward_pre_hook(lambda m, i: cap.__setitem__("seg_in", npy(i[0])))
    with torch.no_grad():
        f2, s2 = m0(t(ir), t(vis))
        seg_in_b2 = cap["seg_in"].copy()
        f1, s1 = m0(t(ir[:1]), t(vis[:1]))
        seg_in_b1 = cap["seg_in"].copy()
    h.remove()
    save("gd_colour_glue", ycc=npy(ycc), rgb=npy(rgb), seg_in_b2=seg_in_b2, seg_in_b1=seg_in_b1,
         logits_b2=npy(s2), logits_b1=npy(s1))

    # ---- G-e: WeTr mit_b0 / mit_b3 at 64x96: stage outputs + logits -------------------------
    for bb in ("mit_b0", "mit_b3"):
        m = m0 if bb == "mit_b0" else build_model(R, bb, head=S.head_tag("mit_b3", 4, 64, 96))
        x = t(seg_in_b2)
        xg = x.clone().requires_grad_(True)
        feats = m.denoise_net.encoder(xg)
        logits = m.denoise_net.decoder(feats)
        rr = S.make_feature(41, tuple(logits.shape))
        (logits * t(rr)).sum().backward()
        save("ge_wetr_%s" % bb, c1=npy(feats[0]), c2=npy(feats[1]), c3=npy(feats[2]), c4=npy(feats[3]),
             logits=npy(logits), dx=npy(xg.grad))
        if bb == "mit_b3":
            m3 = m

    # ---- G-i: state_dict layouts -------------------------------------------------------------
    for bb, m in (("mit_b0", m0), ("mit_b3", m3)):
        layout = {k: list(v.shape) for k, v in m.state_dict().items()}
        with open(os.path.join(OUT, "gi_state_dict_%s.json" % bb), "w") as f:
            json.dump(layout, f, indent=0, sort_keys=True)
    mc = build_model(R, "mit_b0", cls="Network_MM_CompModel")
    assert list(mc.state_dict().keys()) == list(m0.state_dict().keys())

    # ---- G-f: full model, config 1 (4 x 64x96 here; 4 x 480x640 is timed, not stored) and
    #           ONE 480x640 pair through mit_b3 -------------------------------------------------
    ir, vis, lab = S.make_batch(4, 64, 96)
    with torch.no_grad():
        f, s = m3(t(ir), t(vis))
        up = torch.nn.functional.interpolate(s, size=lab.shape[1:], mode="bilinear", align_corners=False)
        pred = up.argmax(1)
    from oracle.paif_oracle import confusion_matrix

    conf = confusion_matrix(lab, npy(pred))
    prec, rec, iou = R["util"].compute_results(conf)
    print("gf_model_b3_4x64x96: class shares of the reference's prediction", class_share(npy(pred)), "mIoU %.4f" % np.nanmean(iou))
    save("gf_model_b3_4x64x96", fused=npy(f), logits=npy(s), pred=npy(pred).astype(np.uint8), conf=conf,
         precision=prec, recall=rec, iou=iou, class_share=class_share(npy(pred)))
    S.load_formula_weights(m3, head=S.head_tag("mit_b3", 1, 480, 640))       # the head fitted on this very pair
    ir, vis, lab = S.make_batch(1, 480, 640)
    with torch.no_grad():
        f, s = m3(t(ir), t(vis))
        up = torch.nn.functional.interpolate(s, size=lab.shape[1:], mode="bilinear", align_corners=False)
        pred = up.argmax(1)
    # the same pair through the reference in float64: documents the reference's own fp32 noise floor at
    # full size (its cumsum box filter loses ~1e-4..1e-3 towards the far image edges)
    m3.double()
    torch.set_default_dtype(torch.float64)  # YCrCb2RGB builds its matrix with torch.tensor(...) (:96-100)
    with torch.no_grad():
        f64, s64 = m3(t(ir).double(), t(vis).double())
    torch.set_default_dtype(torch.float32)
    m3.float()
    up64 = torch.nn.functional.interpolate(s64, size=lab.shape[1:], mode="bilinear", align_corners=False)
    pred64 = up64.argmax(1)
    conf = confusion_matrix(lab, npy(pred))
    conf64 = confusion_matrix(lab, npy(pred64))
    print("gf_model_b3_1x480x640: class shares", class_share(npy(pred)), "mIoU %.5f (float64 run %.5f), pixels the reference's "
          "float32 and float64 runs disagree on: %d" % (np.nanmean(R["util"].compute_results(conf)[2]),
                                                        np.nanmean(R["util"].compute_results(conf64)[2]), int((npy(pred) != npy(pred64)).sum())))
    save("gf_model_b3_1x480x640", fused=npy(f).astype(np.float32), logits=npy(s),
         pred=npy(pred).astype(np.uint8), fused64=npy(f64).astype(np.float32), logits64=npy(s64).astype(np.float32),
         pred64=npy(pred64).astype(np.uint8), conf=conf, conf64=conf64, class_share=class_share(npy(pred)))

    # ---- G-g: attack_both (3 iters, mit_b0, 2 x 64x96), PGD / segPGD / cosPGD ------------------
    ir, vis, lab = S.make_batch(2, 64, 96)
    eps, alpha = 8 / 255.0, 2 / 255.0
    for way in ("PGD", "segPGD", "cosPGD"):
        # the reference draws delta0 from the global RNG (attack/attack.py:434,439): pin the seed,
        # replay the same draws here and store them so both sides start from the same point.
        torch.manual_seed(1234)
        d0_ir = torch.zeros_like(t(ir)).uniform_(-eps, eps)
        d0_vis = torch.zeros_like(t(vis)).uniform_(-eps, eps)
        rec_maps = []

        def recording_model(a, b, _m=m0, _rec=rec_maps):
            fz, sg = _m(a, b)
            _rec.append(sg.detach().clone())
            return fz, sg

        for p_ in m0.parameters():
            p_.grad = None
        torch.manual_seed(1234)
        with torch.no_grad():
            d_ir, d_vis = R["attack"].attack_both(recording_model, t(vis), t(ir), t(lab), epsilon=eps, alpha=alpha,
                                                  attack_iters=3, attack_loss="l_seg", attack_way=way)
        crit = R["attack"].Seg_loss()
        losses = []
        for i, sg in enumerate(rec_maps):
            outp = torch.nn.functional.interpolate(sg, size=lab.shape[1:], mode="bilinear", align_corners=False)
            losses.append(float(O.attack_loss_value(outp, t(lab), way, i, 3)))
            if way == "PGD":
                assert abs(losses[-1] - float(crit(outp, t(lab)))) < 1e-7
        save("gg_attack_%s" % way, d0_ir=npy(d0_ir), d0_vis=npy(d0_vis), delta_ir=npy(d_ir), delta_vis=npy(d_vis),
             gsum_sign_ir=np.sign(npy(d_ir.grad)).astype(np.int8), gsum_sign_vis=np.sign(npy(d_vis.grad)).astype(np.int8),
             gsum_ir=npy(d_ir.grad), gsum_vis=npy(d_vis.grad), losses=np.array(losses))

    # ---- G-g2: the single-modality attacks nobody calls (attack/attack.py:117-411), 2 iterations ------------
    with torch.no_grad():
        X_fusion = m0(t(ir), t(vis))[0]
    out = {}
    cases = [
        ("pgd_attack_ir.l_seg", lambda: R["attack"].pgd_attack_ir(m0, t(vis), t(ir), X_fusion, t(lab), eps, alpha, 2, 1, "l_seg"), ir.shape),
        ("pgd_attack_ir.l_2", lambda: R["attack"].pgd_attack_ir(m0, t(vis), t(ir), X_fusion, t(lab), eps, alpha, 2, 1, "l_2"), ir.shape),
        ("pgd_attack_vision.l_seg", lambda: R["attack"].pgd_attack_vision(m0, t(vis), t(ir), X_fusion, t(lab), eps, alpha, 2, 1, "l_seg"), vis.shape),
        # fgsm_ir (:247-304) cannot run in the reference: with_mask=False hits UnboundLocalError on `black_X` (:295),
        # with_mask=True calls get_ir_mask whose map_generate3 is undefined (:232-244) -> no golden, not built
        ("seg_pgd.vis", lambda: R["attack"].seg_pgd(m0, t(vis), t(ir), X_fusion, t(lab), eps, alpha, 2, 1, "l_seg", "vis"), vis.shape),
        ("cos_pgd.ir", lambda: R["attack"].cos_pgd(m0, t(vis), t(ir), X_fusion, t(lab), eps, alpha, 2, 1, "l_seg", "ir"), ir.shape),
    ]
    for name, fn, shp in cases:
        torch.manual_seed(4321)
        d0 = torch.zeros(shp).uniform_(-eps, eps)
        torch.manual_seed(4321)
        with ref_import.quiet():
            d = fn()
        out[name + ".d0"] = npy(d0)
        out[name + ".delta"] = npy(d)
    save("gg2_single_modality_attacks", **out)

    # ---- G-h: losses, metrics, schedule --------------------------------------------------------
    logits = S.make_feature(51, (2, 9, 24, 32), -3, 3)
    lab_s = S.make_label(3, 24, 32)[None].repeat(2, 0)
    lg = t(logits).requires_grad_(True)
    l_seg = R["attack"].Seg_loss()(lg, t(lab_s))
    l_seg.backward()
    a = S.make_smooth_feature(52, 2, 1, 32, 40)
    b = S.make_smooth_feature(53, 2, 1, 32, 40)
    l_ssim = R["ssim"].ssim(t(a), t(b))
    l_fus = R["loss"].Fusionloss_grad2()(t(a), t(a), t(a), t(b))
    conf = np.arange(81).reshape(9, 9) % 7
    conf[:, 4] = 0
    conf[4, :] = 0  # an empty class -> NaN
    prec, rec, iou = R["util"].compute_results(conf)
    opt = R["optimizer"].PolyWarmupAdamW([torch.nn.Parameter(torch.zeros(1))], lr=8e-5, weight_decay=0.01,
                                         betas=(0.9, 0.999), warmup_iter=3000, max_iter=160000, warmup_ratio=1e-5,
                                         power=1.0)
    lrs = []
    for step in (0, 1, 2999, 3000, 80000, 159999):
        opt.global_step = step
        opt.param_groups[0]["params"][0].grad = torch.zeros(1)
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
    save("gh_losses_metrics", seg_loss=np.array(float(l_seg)), seg_loss_dlogits=npy(lg.grad),
         ssim=np.array(float(l_ssim)), fusionloss_grad2=np.array(float(l_fus)), conf=conf, precision=prec,
         recall=rec, iou=iou, lr_steps=np.array([0, 1, 2999, 3000, 80000, 159999]), lrs=np.array(lrs))
    print("done")


if __name__ == "__main__":
    main()
