"""GPU parity IN THE ARITHMETIC AND AT THE BATCH SIZES bench.py RUNS.

The gradient / trajectory tests of test_attack_gpu.py and test_backward_fullsize_gpu.py pin the exact-fp32 kernels
(`set_conv_precision("f32")`, `set_gemm_precision("f32")`).  Until round 3 the product default -- and what `bench.py --workload
pgd|train` timed -- was split-bf16 conv products (forward AND dgrad), split-bf16 attention forward/backward and "auto" GEMMs inside
the attack loop as well.  Every test here is parametrised over precision in {exact, default, fast} (see the fixture); the numbers
were MEASURED on MI355X (each test writes them to gpurun_out/parity_metrics.json; the round-3 record is committed as
profiles/r03_parity_default_arithmetic.json) and are stated next to the reference arithmetic's own
float32-vs-float64 disagreement on the same quantity (tests/golden/gn_attack_PGD10.npz: `floor_*`), which is the yardstick: a PGD
trajectory is chaotic in sign(g), so "as close to the float64 run as the reference's own float32 run is" is the strongest
statement available for either arithmetic.

Also here: the fusion network at the benchmarked batch (configs[1] B=8, configs[2] B=16, 480x640) -- every other full-size test
is B=1, and the persistent kernels' tile ranges straddle image boundaries only when B>1 at full size.
"""
import json
import os

import numpy as np
import pytest
import torch

from paif_amd import ops, synthetic as S
from tests import helpers as Hh
from tests.helpers import t, maxabs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS, ALPHA = 8 / 255., 2 / 255.


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _record(name, **metrics):
    """Measured parity numbers -> gpurun_out/parity_metrics.json (scratch; the stated bounds live in the asserts)."""
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d):
        return
    path = os.path.join(d, "parity_metrics.json")
    try:
        allm = json.load(open(path))
    except (OSError, ValueError):
        allm = {}
    allm[name] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in metrics.items()}
    json.dump(allm, open(path, "w"), indent=1, sort_keys=True)


@pytest.fixture(params=["exact", "default", "fast"])
def precision(request):
    """exact   = every kernel fp32-exact (what test_attack_gpu.py / test_backward_fullsize_gpu.py pin);
    default = what ops.CONFIG ships with and bench.py runs: split-bf16 convs / "auto" GEMMs for plain forwards and backwards, and
              -- since the round-3 finding below -- the ATTACK LOOPS on fp32-level arithmetic: convs as three-piece bf16 splits (six
              MFMAs per product, 2^-25), GEMMs / attention exact (CONFIG["attack_precision"] = "bf16x6");
    fast    = ops.set_attack_precision("fast"): the attack loops in split-bf16 too (bench.py --attack-precision fast)."""
    old = dict(ops.CONFIG)
    if request.param == "exact":
        ops.set_conv_precision("f32")
        ops.set_gemm_precision("f32")
        ops.set_attack_precision("exact")
    else:
        ops.set_conv_precision("bf16x3")
        ops.set_gemm_precision("auto")
        ops.set_attack_precision("bf16x6" if request.param == "default" else "fast")
    yield request.param
    ops.CONFIG.update(old)


def _model(bb="mit_b0", head="cal"):
    """head: "cal" = the calibrated segmentation head of the 64x96 cases (tests/helpers.py HEAD64), or an explicit tag (HEAD480)."""
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT

    m = Network_MM_Searched(32, FUSION_AT, None, None, bb, num_classes=9).eval()
    S.load_formula_weights(m, head=Hh.HEAD64[bb] if head == "cal" else head)
    return m.to(_dev())


def _sign_mismatch(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else b
    return float((np.sign(a) != np.sign(b)).mean())


# ------------------------------------------------------------------------------------------------------------------
# A1: PGD trajectories
# ------------------------------------------------------------------------------------------------------------------
def test_attack_both_pgd3_trajectory(golden, precision):
    """attack/attack.py:443-512, 3 iterations, vs the reference's own trace (gg_attack_PGD).  exact: loss rel. 1e-4, sign
    mismatch / differing delta <= 2e-3.  default (measured): see the bounds below."""
    from paif_amd.attack.attack import attack_both

    g = golden("gg_attack_PGD")
    m = _model("mit_b0")
    ir, vis, lab = S.make_batch(2, 64, 96)
    trace = []
    with torch.no_grad():
        d_ir, d_vis = attack_both(m, t(vis).to(_dev()), t(ir).to(_dev()), t(lab).to(_dev()), epsilon=EPS, alpha=ALPHA, attack_iters=3,
                                  attack_loss='l_seg', attack_way='PGD', delta0_ir=t(g["d0_ir"]), delta0_vis=t(g["d0_vis"]), trace=trace)
    losses = np.array([s["loss"] for s in trace])
    loss_rel = np.abs(losses - g["losses"]) / np.abs(g["losses"])
    sm = [_sign_mismatch(trace[-1]["g_ir"], g["gsum_ir"]), _sign_mismatch(trace[-1]["g_vis"], g["gsum_vis"])]
    dm = [float((np.abs(d_ir.detach().cpu().numpy() - g["delta_ir"]) > 1e-6).mean()), float((np.abs(d_vis.detach().cpu().numpy() - g["delta_vis"]) > 1e-6).mean())]
    _record("pgd3_2x64x96_mit_b0[%s]" % precision, loss_rel=loss_rel, sign_mismatch=sm, delta_mismatch=dm)
    lim = BOUNDS["pgd3"][precision]
    assert loss_rel.max() <= lim["loss"], loss_rel
    assert max(sm) <= lim["sign"], sm
    assert max(dm) <= lim["delta"], dm
    assert float(d_ir.detach().abs().max()) <= EPS + 1e-7 and float(d_vis.detach().abs().max()) <= EPS + 1e-7


def test_attack_both_pgd10_trajectory_and_attacked_miou(golden, precision, tag=None):
    """H2 at the iteration count configs[3] uses: TEN accumulated-gradient iterations (the reference never zeroes delta.grad,
    attack/attack.py:501-512), then the harness forward on the attacked pair (robust_test.py:143-166), vs the reference's own
    run (gn_attack_PGD10).  Stated per iteration against the reference's float32-vs-float64 disagreement (`floor_*`)."""
    from paif_amd.attack.attack import attack_both
    from paif_amd.util.util import ConfusionMeter, compute_results

    g = golden("gn_attack_PGD10")
    m = _model("mit_b0")
    ir, vis, lab = S.make_batch(2, 64, 96)
    dev = _dev()
    irt, vist, labt = t(ir).to(dev), t(vis).to(dev), t(lab).to(dev)
    trace = []
    with torch.no_grad():
        d_ir, d_vis = attack_both(m, vist, irt, labt, epsilon=EPS, alpha=ALPHA, attack_iters=10, attack_loss='l_seg', attack_way='PGD',
                                  delta0_ir=t(g["d0_ir"]), delta0_vis=t(g["d0_vis"]), trace=trace)
        fused, seg = m(ops.add(irt, d_ir.detach()), ops.add(vist, d_vis.detach()))
        meter = ConfusionMeter(9, dev)
        pred = meter.update(seg, labt)
    losses = np.array([s["loss"] for s in trace])
    loss_rel = np.abs(losses - g["losses"]) / np.abs(g["losses"])
    # per-iteration sign mismatch of the running gradient sum vs the reference's float32 run AND vs its float64 run
    sm32 = np.array([[_sign_mismatch(np.sign(s["g_ir"].cpu().numpy()), g["sign_ir_per_iter"][i]),
                      _sign_mismatch(np.sign(s["g_vis"].cpu().numpy()), g["sign_vis_per_iter"][i])] for i, s in enumerate(trace)])
    sm64 = np.array([[_sign_mismatch(np.sign(s["g_ir"].cpu().numpy()), g["sign64_ir_per_iter"][i]),
                      _sign_mismatch(np.sign(s["g_vis"].cpu().numpy()), g["sign64_vis_per_iter"][i])] for i, s in enumerate(trace)])
    dm32 = [float((np.abs(d_ir.detach().cpu().numpy() - g["delta_ir"]) > 1e-6).mean()), float((np.abs(d_vis.detach().cpu().numpy() - g["delta_vis"]) > 1e-6).mean())]
    dm64 = [float((np.abs(d_ir.detach().cpu().numpy() - g["delta64_ir"]) > 1e-6).mean()), float((np.abs(d_vis.detach().cpu().numpy() - g["delta64_vis"]) > 1e-6).mean())]
    moved = int((pred.cpu().numpy() != g["pred"]).sum())
    miou = float(np.mean(np.nan_to_num(compute_results(meter.conf.cpu().numpy())[2])))
    miou_ref = float(np.mean(np.nan_to_num(compute_results(g["conf"])[2])))
    # the golden is discriminating: the reference's CLEAN map has all 9 classes and mIoU 0.323, its attacked map 3 classes and
    # mIoU 0.035 (the attack moves 83 % of the pixels); its own float32 and float64 runs disagree on 2 attacked pixels
    Hh.assert_multiclass(g["pred_clean"], min_classes=9)
    Hh.assert_multiclass(g["pred"], min_classes=3)
    miou_clean_ref = float(np.mean(np.nan_to_num(compute_results(g["conf_clean"])[2])))
    assert miou_clean_ref - miou_ref >= 0.2, (miou_clean_ref, miou_ref)
    _record("pgd10_2x64x96_mit_b0[%s]" % (tag or precision), loss_rel=loss_rel, sign_mismatch_vs_ref32=sm32, sign_mismatch_vs_ref64=sm64,
            delta_mismatch_vs_ref32=dm32, delta_mismatch_vs_ref64=dm64, moved_pixels=moved, miou=miou, miou_ref=miou_ref,
            miou_clean_ref=miou_clean_ref, ref_moved_pixels_f32_vs_f64=int((g["pred"] != g["pred64"]).sum()),
            ref_floor_sign=g["floor_sign"], ref_floor_loss=g["floor_loss"], ref_floor_delta=g["floor_delta"])
    lim = BOUNDS["pgd10"][precision]
    assert abs(miou - miou_ref) <= 1e-3, (miou, miou_ref)                      # mIoU within 0.1 pt, every arithmetic
    assert moved <= lim["moved"] * pred.numel(), moved
    if precision == "fast":
        # THE FINDING (round 3): with split-bf16 conv products inside the attack loop the trajectory leaves the reference's:
        # measured sign mismatch 2.4e-4 (iteration 1) -> 9e-4 (3) -> 3.5e-3 (5) -> 2.5e-2 (10), differing delta 5.3 %, loss
        # 3.3e-3 -- 60x the reference's own float32-vs-float64 disagreement (floor_sign <= 4.1e-4).  Not a tolerance: the
        # product default is "exact" because of it; what is asserted here is the measured envelope (x2) so that a
        # regression of the fast mode is still caught, and that the A1 bound holds for the first three iterations.
        assert (sm64[:3] <= 2e-3).all() and loss_rel[:3].max() <= 1e-4, (sm64, loss_rel)
        assert sm64.max() <= 5e-2 and max(dm64) <= 0.11 and loss_rel.max() <= 1e-2, (sm64, dm64, loss_rel)
        return
    assert loss_rel.max() <= lim["loss"], loss_rel
    # the yardstick: not farther from the float64 trajectory than `k` x the reference's own float32 run is (+ an absolute 1e-3):
    # SURVEY 8(a) A1's <= 1e-3 per iteration on top of the reference arithmetic's own floor
    if tag == "f16x3_forward_and_backward":
        # fp16 pairs in the REVERSE pass, forced onto this small case: the guided filter's reverse pass multiplies its input's error by
        # up to 1 / (var + 1e-4), and with the pairs' error (relative to a sum's largest term, not to each element) ~70 % of the
        # starting points leave the exact kernels' trajectory by iteration 10 -- 11 of 16 with the round-1 guided-filter kernels, 12
        # of 16 with the round-6 streaming pair, sign mismatch 1e-4 ... 1.4e-2 at iteration 10 (tools/pgd_seed_study.py,
        # profiles/r06_pgd_seed_study_*.json).  The golden start is one of the quiet ones under the round-1 kernels (0 in all ten
        # iterations) and one that departs in iteration 3 under the streaming pair (3e-5, 8e-5, 3e-4, 6e-4, 9e-4, 1.4e-3, 2.2e-3,
        # 2.4e-3).  Asserted: A1 over the first seven iterations, the measured envelope (x2) behind them -- and, above, the attacked
        # mIoU and the moved pixels, which do not feel any of it (2 pixels, the reference's own fp32-vs-fp64 figure).
        assert (sm64[:7] <= lim["sign_k"] * g["floor_sign"][:7] + 1e-3).all(), (sm64, g["floor_sign"])
        assert sm64.max() <= 5e-3 and max(dm64) <= 1.5e-2, (sm64, dm64)
        return
    assert (sm64 <= lim["sign_k"] * g["floor_sign"] + 1e-3).all(), (sm64, g["floor_sign"])
    assert max(dm64) <= lim["sign_k"] * float(g["floor_delta"].max()) + 1e-3, (dm64, g["floor_delta"])
    assert float(d_ir.detach().abs().max()) <= EPS + 1e-7 and float(d_vis.detach().abs().max()) <= EPS + 1e-7


@pytest.mark.parametrize("mode", ["x6_everywhere", "f16x3_forward", "f16x3_forward_and_backward"])
def test_pgd10_gate_with_the_split_gemms_forced_at_this_size(golden, mode):
    """The "auto6" rule sends a GEMM to the split kernels only from 2,048 rows up: at 2x64x96 (768 tokens at most) the gate above runs
    every GEMM on the exact fp32 MFMA and says nothing about the arithmetic configs[3] actually uses at 480x640.  Here the threshold is
    lowered to 1 row, and the SAME gate (the bounds of "default") is run with
      x6_everywhere               three-piece bf16 convs / K >= 256 GEMMs in both passes (round 4's attack arithmetic),
      f16x3_forward               fp16 pairs (two 11-bit pieces, three MFMAs) in the forward pass, three-piece bf16 in the reverse pass,
      f16x3_forward_and_backward  fp16 pairs in both, the reverse pass scaled by ops.attack_grad_scale (an exact power of two)."""
    old = dict(ops.CONFIG)
    try:
        ops.set_conv_precision("bf16x3")
        ops.set_gemm_precision("auto")
        ops.set_attack_precision("bf16x6")
        ops.CONFIG["gemm_split_min_m"] = 1
        ops.CONFIG["attack_fwd_f16x3"] = mode != "x6_everywhere"
        ops.CONFIG["attack_bwd_f16x3"] = mode == "f16x3_forward_and_backward"
        ops.CONFIG["infer_f16x3"] = mode != "x6_everywhere"        # (the harness forward on the attacked pair after the loop)
        timer = ops.KernelTimer(lambda tag: tag.startswith("gemm_mfma") or tag.startswith("conv_mfma"))
        ops.TIMER = timer
        try:
            test_attack_both_pgd10_trajectory_and_attacked_miou(golden, "default", tag=mode)
        finally:
            ops.TIMER = None
        torch.cuda.synchronize()
        seen = set(timer.summary())
        x6 = {t for t in seen if t == "gemm_mfma_bf16x6" or t.startswith("conv_mfma_bf16x6<")}
        h3 = {t for t in seen if t == "gemm_mfma_f16x3" or t.startswith("conv_mfma_f16x3<")}
        if mode == "x6_everywhere":
            assert x6 and not h3, seen
        elif mode == "f16x3_forward":
            assert "gemm_mfma_f16x3" in h3 and "gemm_mfma_bf16x6" in x6 and len(h3) > 1 and len(x6) > 1, seen
        else:                                   # the three-piece form is left only where a kernel has no fp16-pair form
            assert "gemm_mfma_f16x3" in h3 and len(h3) > 1, seen
    finally:
        ops.CONFIG.clear()
        ops.CONFIG.update(old)


def test_one_pgd_iteration_at_480x640_mit_b3(precision):
    """One full attack_both iteration at the configs[3] shape class (1x480x640, mit_b3) vs the oracle's autograd on the host."""
    from oracle import paif_oracle as O
    from paif_amd.attack.attack import attack_both

    m = _model("mit_b3", head=Hh.HEAD480)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ir, vis, lab = S.make_batch(1, 480, 640)
    d0i = t(S.make_delta0(5, ir.shape, EPS))
    d0v = t(S.make_delta0(105, vis.shape, EPS))
    trace = []
    with torch.no_grad():
        d_ir, d_vis = attack_both(m, t(vis).to(_dev()), t(ir).to(_dev()), t(lab).to(_dev()), epsilon=EPS, alpha=ALPHA, attack_iters=1,
                                  attack_loss='l_seg', attack_way='PGD', delta0_ir=d0i, delta0_vis=d0v, trace=trace)
    key = "_pgd1_oracle"
    if key not in _CACHE:                                     # the host run is the same for both precisions: once per session
        otrace = []
        od = O.attack_both(lambda a, b: O.model_forward(a, b, sd, "mit_b3"), t(vis), t(ir), t(lab), d0i, d0v, epsilon=EPS, alpha=ALPHA,
                           attack_iters=1, attack_way="PGD", trace=otrace)
        _CACHE[key] = (otrace, od)
    otrace, (od_ir, od_vis) = _CACHE[key]
    loss_rel = abs(trace[0]["loss"] - otrace[0]["loss"]) / abs(otrace[0]["loss"])
    sm, mx, mean = [], [], []
    for mine, ref in ((trace[0]["g_ir"], otrace[0]["g_ir"]), (trace[0]["g_vis"], otrace[0]["g_vis"])):
        a, b = mine.cpu().numpy(), ref.numpy()
        sm.append(_sign_mismatch(a, b))
        mx.append(float(np.abs(a - b).max() / np.abs(b).max()))
        mean.append(float(np.abs(a - b).mean() / np.abs(b).max()))
    dm = [float((np.abs(d_ir.detach().cpu().numpy() - od_ir.numpy()) > 1e-6).mean()), float((np.abs(d_vis.detach().cpu().numpy() - od_vis.numpy()) > 1e-6).mean())]
    _record("pgd1_1x480x640_mit_b3[%s]" % precision, loss_rel=loss_rel, sign_mismatch=sm, grad_max_rel=mx, grad_mean_rel=mean, delta_mismatch=dm)
    lim = BOUNDS["pgd1_full"][precision]
    assert loss_rel <= lim["loss"], loss_rel
    assert max(sm) <= lim["sign"] and max(dm) <= lim["sign"], (sm, dm)
    assert max(mx) <= lim["gmax"] and max(mean) <= lim["gmean"], (mx, mean)


@pytest.mark.parametrize("bb", ["mit_b0", "mit_b3"])
def test_wetr_input_grad(golden, precision, bb):
    g = golden("ge_wetr_" + bb)
    m = _model(bb)
    x = t(golden("gd_colour_glue")["seg_in_b2"]).to(_dev()).requires_grad_(True)
    logits = m.denoise_net(x)
    (logits * t(S.make_feature(41, tuple(logits.shape))).to(_dev())).sum().backward()
    err = maxabs(x.grad.cpu(), g["dx"]) / max(1.0, float(np.abs(g["dx"]).max()))
    lerr = maxabs(logits.detach().cpu(), g["logits"]) / max(1.0, float(np.abs(g["logits"]).max()))
    _record("wetr_input_grad_%s[%s]" % (bb, precision), dx_rel=err, logits_rel=lerr)
    assert lerr <= 1e-4
    assert err <= BOUNDS["wetr_dx"][precision], err


def test_harness_pgd_eval(precision):
    """H2 (robust_test.py:95-239) through the harness: PGD-3 on 2 pairs (mit_b0), attacked mIoU vs the CPU oracle's run."""
    from oracle import paif_oracle as O
    from paif_amd.harness import val_segformer_robust

    m = _model("mit_b0")
    sd = Hh.model_sd("mit_b0", Hh.HEAD64["mit_b0"])
    ir, vis, lab = S.make_batch(2, 64, 96)
    d0_ir = t(S.make_delta0(0, ir.shape, EPS))
    d0_vis = t(S.make_delta0(1, vis.shape, EPS))
    out = val_segformer_robust(m, [(t(vis).to(_dev()), t(ir).to(_dev()), t(lab).to(_dev()))], attack_iters=3,
                               delta0=lambda bi, a, b: (d0_ir, d0_vis))
    if "_harness" not in _CACHE:
        fwd = lambda a, b: O.model_forward(a, b, sd, "mit_b0")
        o_ir, o_vis = O.attack_both(fwd, t(vis), t(ir), t(lab), d0_ir, d0_vis, EPS, ALPHA, 3, "PGD")
        with torch.no_grad():
            _, seg = fwd(t(ir) + o_ir, t(vis) + o_vis)
            up = torch.nn.functional.interpolate(seg, size=lab.shape[1:], mode="bilinear", align_corners=False)
        Hh.assert_multiclass(up.argmax(1).numpy(), min_classes=2)
        _CACHE["_harness"] = O.confusion_matrix(lab, up.argmax(1).numpy())
    conf = _CACHE["_harness"]
    iou = O.compute_results(conf)[2]
    dmiou = abs(out["miou"] - float(np.mean(np.nan_to_num(iou))))
    moved = float(np.abs(out["conf"] - conf).sum() / conf.sum())
    _record("harness_pgd3[%s]" % precision, dmiou=dmiou, moved_frac=moved)
    assert dmiou <= 1e-3                                                         # mIoU within 0.1 pt
    # <= 0.2 % of the pixels move (exact / default: none).  "fast" (opt-in: split-bf16 products inside the loop, a trajectory that is
    # KNOWN to leave the reference's, see the PGD-10 test) is held to its measured envelope as there: 0.17-0.23 % over the guided-filter
    # forms of round 6, bound 0.5 %
    assert moved <= (0.005 if precision == "fast" else 0.002), moved


_CACHE = {}

# Stated bounds.  `exact` = the SURVEY 8(a) A1 metric as round 2 stated it.  `default` = measured on MI355X in round 3
# (profiles/r03_parity_default_arithmetic.json holds the numbers), with head-room of ~2x over the measurement; the comparison
# with the reference's own float32-vs-float64 floor is asserted inside the PGD-10 test.
_A1 = dict(loss=1e-4, sign=2e-3, delta=2e-3)
BOUNDS = {
    # measured (fast): loss 2.6e-5, sign 9e-4, delta 1.5e-3
    "pgd3": {"exact": _A1, "default": _A1, "fast": dict(loss=1e-4, sign=2e-3, delta=3e-3)},
    # measured (exact / default): loss 6.3e-5 (the reference's own float32 run: 6.0e-5), sign vs float64 <= 8.1e-5, delta 8.1e-5
    "pgd10": {"exact": dict(loss=2e-4, sign_k=1.5, moved=1e-3), "default": dict(loss=2e-4, sign_k=1.5, moved=1e-3), "fast": dict(moved=5e-3)},      # fast: 32 of 12,288 attacked pixels land elsewhere (2.6e-3); default / exact: 2 -- the reference's own float32-vs-float64 count
    # one iteration at 480x640 (vs the oracle's float32 autograd, itself ~2e-4 from its float64): measured sign 2.1e-4 / 2.6e-4 (fast)
    "pgd1_full": {k: dict(loss=1e-4, sign=2e-3, gmax=5e-2, gmean=2e-3) for k in ("exact", "default", "fast")},
    # plain autograd through the SegFormer in the default arithmetic: measured 2.6e-6 (exact 5e-7)
    "wetr_dx": {"exact": 1e-4, "default": 1e-4, "fast": 1e-4},
    # bench.py's default configuration (B=8, 480x640, bf16 storage): measured on MI355X in round 4 (+20 %)
    # measured: bf16 fused 1.35e-2 / 7.8e-4, uint8 image max 8 grey levels, 5.4 % of the values > 1 level, 12.4 % != 0;
    #           bf16_split 1.46e-2 / 6.4e-4, max 9 levels, 5.2 % > 1 level
    # f16 (bench.py's default since round 5): fused 1.8e-3 / 6.3e-5 vs float64; the uint8 image: 99.5 % of the values identical to the
    # reference's, max 3 grey levels (bf16: 88 % identical, max 9)
    "b8_bf16": {"f16": dict(fused_max=2.5e-3, fused_mean=1.0e-4, lev_max=4, lev_gt1=0.008),
                "bf16": dict(fused_max=1.7e-2, fused_mean=9.5e-4, lev_max=10, lev_gt1=0.065),
                "bf16_split": dict(fused_max=1.8e-2, fused_mean=8e-4, lev_max=11, lev_gt1=0.063)},
    # configs[2] at B=16 vs the oracle on the host.  measured: f32 fused 1.8e-4, logits 4.9e-5 of the range, 408 of 4,915,200 pixels
    # (agreement 0.99992; each decided by the oracle by < 2.2e-5 of the range); bf16 fused 2.2e-2, logits 8.3e-3, agreement 0.99518
    # f16 (16 samples = 4.9 M pixels): SURVEY 8(d)'s 99.9 % asserted on the batch
    "b16": {"f32": dict(fused=2.2e-4, logits=1e-4, agree=0.9998), "f16": dict(fused=3e-3, logits=1.5e-3, agree=0.999),
            "bf16": dict(fused=2.6e-2, logits=1.2e-2, agree=0.985)},
}


# ------------------------------------------------------------------------------------------------------------------
# the benchmarked batch sizes at 480x640
# ------------------------------------------------------------------------------------------------------------------
def test_fusion_b8_480x640_samplewise_equals_b1_and_golden(golden):
    """configs[1] as bench.py runs it (B=8, 480x640, default arithmetic): the fusion network is per-sample, so sample i of the
    B=8 launch must equal the B=1 launch of sample i BIT FOR BIT (same kernels, same per-sample arithmetic; only the persistent
    kernels' tile ranges and the tile->workgroup map differ), and sample 0 must sit on the reference's golden."""
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    from paif_amd.genotypes import FUSION_AT

    dev = _dev()
    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    # the weights the fusion net has INSIDE the composite model's golden (the formula is keyed on the full state_dict key)
    net.load_state_dict({k: t(S.formula_tensor("enhance_net." + k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()},
                        strict=True)
    net.to(dev)
    ir, vis, _ = S.make_batch(8, 480, 640)
    irt, vist = t(ir).to(dev), t(vis).to(dev)
    with torch.no_grad():
        ycc = ops.rgb2ycrcb(vist)
        f8 = net(irt, ycc).clone()
        worst = 0.0
        for i in (0, 3, 7):
            f1 = net(irt[i:i + 1].contiguous(), ops.rgb2ycrcb(vist[i:i + 1].contiguous()))
            worst = max(worst, float((f8[i:i + 1] - f1).abs().max()))
    g = golden("gf_model_b3_1x480x640")
    e_gold = maxabs(f8[0:1].cpu(), g["fused"])
    e64 = maxabs(f8[0:1].cpu(), g["fused64"])
    floor = maxabs(g["fused"], g["fused64"])                  # the reference's own float32-vs-float64 distance at this size
    _record("fusion_b8_480x640", samplewise_max=worst, vs_golden=e_gold, vs_fp64=e64, ref_floor=floor)
    assert worst <= 1e-6, worst
    assert e64 <= floor, (e64, floor)                         # not farther from the float64 result than the reference's own run
    assert e_gold <= 2 * floor + 1e-5


def _fusion_net_in_model():
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    from paif_amd.genotypes import FUSION_AT

    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    # the weights the fusion net has INSIDE the composite model's golden (the formula is keyed on the full state_dict key)
    net.load_state_dict({k: t(S.formula_tensor("enhance_net." + k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()},
                        strict=True)
    return net.to(_dev())


@pytest.mark.parametrize("mode", ["f16", "bf16", "bf16_split"])
def test_fusion_b8_480x640_bf16_storage_is_what_bench_runs(golden, mode):
    """THE BENCHMARKED CONFIGURATION END TO END: `bench.py` default = fusion forward, B=8, 480x640,
    `--storage bf16`.  (a) sample i of the B=8 launch equals the B=1 launch of sample i bit for bit in the same storage mode;
    (b) sample 0 against the reference's float64 fused plane: max / mean |d| bounded by the measurement (+20 %);
    (c) the uint8 fused IMAGE the harness writes (test_original.py:181-197) from the bf16 result vs the one the same writer (pinned
    bit-exact by gk_fused_writer) produces from the reference's own fused plane: grey-level difference histogram."""
    from oracle import paif_oracle as O
    from paif_amd import harness

    dev = _dev()
    net = _fusion_net_in_model()
    ir, vis, _ = S.make_batch(8, 480, 640)
    irt, vist = t(ir).to(dev), t(vis).to(dev)
    old = ops.CONFIG["storage"]
    ops.set_storage(mode)
    try:
        with torch.no_grad():
            ycc = ops.rgb2ycrcb(vist)
            f8 = net(irt, ycc).clone()
            worst = 0.0
            for i in (0, 3, 7):
                f1 = net(irt[i:i + 1].contiguous(), ops.rgb2ycrcb(vist[i:i + 1].contiguous()))
                worst = max(worst, float((f8[i:i + 1] - f1).abs().max()))
            img = harness.fused_to_uint8(f8[0:1].contiguous(), vist[0:1].contiguous()).cpu().numpy()
    finally:
        ops.set_storage(old)
    g = golden("gf_model_b3_1x480x640")
    d64 = (f8[0:1].cpu().double() - t(g["fused64"]).double()).abs()
    img_ref = O.fused_image_uint8(t(g["fused"]), t(vis[0:1]))
    img_ref = img_ref.numpy() if torch.is_tensor(img_ref) else np.asarray(img_ref)
    lev = np.abs(img.astype(np.int32) - img_ref.astype(np.int32))
    hist = np.bincount(lev.ravel(), minlength=8)
    rec = dict(samplewise_max=worst, fused_max_vs_fp64=float(d64.max()), fused_mean_vs_fp64=float(d64.mean()),
               grey_level_max=int(lev.max()), grey_level_frac_gt1=float((lev > 1).mean()), grey_level_frac_ne0=float((lev > 0).mean()),
               grey_level_hist=hist[:8].tolist())
    _record("fusion_b8_480x640[%s]" % mode, **rec)
    lim = BOUNDS["b8_bf16"][mode]
    assert worst == 0.0, worst                                 # per-sample network: the batch position changes nothing
    assert rec["fused_max_vs_fp64"] <= lim["fused_max"] and rec["fused_mean_vs_fp64"] <= lim["fused_mean"], rec
    assert rec["grey_level_max"] <= lim["lev_max"] and rec["grey_level_frac_gt1"] <= lim["lev_gt1"], rec


@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_fusion_seg_b16_480x640_vs_oracle(golden, mode):
    """configs[2] as bench.py runs it (B=16, 480x640, mit_b3, default arithmetic; `mode` = fp32 storage, the parity configuration,
    and bf16 storage, `bench.py --workload fusion_seg`'s default) against the CPU oracle on the same 16 pairs (the glue's min/max
    is batch-global, core/model_fusion_auto.py:721-723, so the logits of every sample depend on all 16): fused per sample,
    logits, argmax agreement on the x4-upsampled maps, mIoU on the synthetic labels.  Calibrated head: the oracle's maps are
    multi-class on every sample.  ~1 minute of host time (once per session)."""
    from oracle import paif_oracle as O

    dev = _dev()
    m = _model("mit_b3", head=Hh.HEAD480)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    B = 16
    ir, vis, lab = S.make_batch(B, 480, 640)
    old = ops.CONFIG["storage"]
    ops.set_storage(mode)
    try:
        with torch.no_grad():
            fused, seg = m(t(ir).to(dev), t(vis).to(dev))
            fused, seg = fused.cpu(), seg.cpu()
    finally:
        ops.set_storage(old)
    if "_b16" not in _CACHE:
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        fsd = {k[len("enhance_net."):]: v for k, v in sd.items() if k.startswith("enhance_net.")}
        ssd = {k[len("denoise_net."):]: v for k, v in sd.items() if k.startswith("denoise_net.")}
        with torch.no_grad():
            ycc = O.rgb2ycrcb(t(vis))
            of = torch.cat([O.fusion_forward(t(ir[i:i + 1]), ycc[i:i + 1, 0:1], fsd) for i in range(B)])      # per-sample network
            seg_in = O.seg_input_from_fused(of, ycc)                                                          # batch-global min/max
            ol = torch.cat([O.wetr_forward(seg_in[i:i + 4], ssd, "", "mit_b3") for i in range(0, B, 4)])
        _CACHE["_b16"] = (of, ol)
    of, ol = _CACHE["_b16"]
    up = lambda x: torch.nn.functional.interpolate(x, size=(480, 640), mode="bilinear", align_corners=False)
    upo, upm = up(ol), up(seg)
    po, pm = upo.argmax(1).numpy(), upm.argmax(1).numpy()
    shares = [Hh.assert_multiclass(po[i], min_classes=3, min_share=0.04) for i in range(B)]          # every sample's map is multi-class
    srt = torch.sort(upo, dim=1).values
    margin = (srt[:, -1] - srt[:, -2]).numpy()
    rng = float(ol.max() - ol.min())
    e_f = maxabs(fused, of)
    e_l = maxabs(seg, ol) / rng
    dis = po != pm
    agree = float(1.0 - dis.mean())
    worst_margin = float(margin[dis].max() / rng) if dis.any() else 0.0
    miou_o = float(np.nanmean(O.compute_results(O.confusion_matrix(lab, po))[2]))
    miou_m = float(np.nanmean(O.compute_results(O.confusion_matrix(lab, pm))[2]))
    _record("fusion_seg_b16_480x640[%s]" % mode, fused_max=e_f, logits_max_over_range=e_l, argmax_agree=agree, moved_pixels=int(dis.sum()),
            largest_margin_of_a_moved_pixel_over_range=worst_margin, median_margin_over_range=float(np.median(margin) / rng),
            miou_oracle=miou_o, miou=miou_m, min_classes_per_sample=int(min((s_ >= 0.04).sum() for s_ in shares)))
    lim = BOUNDS["b16"][mode]
    assert e_f <= lim["fused"], e_f             # f32: both sides float32 through the guided filter (reference floor 9.8e-5 vs fp64)
    assert e_l <= lim["logits"], e_l            # f32: SURVEY 8(d) logits max-abs <= 1e-3 of the logit range
    assert agree >= lim["agree"], agree
    assert worst_margin <= 2.0 * e_l + 1e-6, (worst_margin, e_l)     # only pixels decided by less than twice the logit error move
    assert abs(miou_m - miou_o) <= 1e-3, (miou_m, miou_o)           # mIoU within 0.1 pt
