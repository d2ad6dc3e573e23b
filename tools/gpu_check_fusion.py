"""Debug harness (GPU box): run every fusion parity check, print max errors, never stop early."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paif_amd import ops, synthetic as S  # noqa: E402
from tests import helpers as Hh  # noqa: E402
from tests.helpers import t, maxabs  # noqa: E402

dev = torch.device("cuda:0")
G = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))


def run(name, fn):
    try:
        t0 = time.time()
        r = fn()
        torch.cuda.synchronize()
        print("%-40s %s  (%.2fs)" % (name, r, time.time() - t0), flush=True)
    except Exception:
        print("%-40s EXC\n%s" % (name, traceback.format_exc()), flush=True)


def prims():
    from paif_amd.core.model_fusion_auto import MixedOp
    g = G("ga_primitives")
    x = t(S.make_smooth_feature(11, 1, 32, 24, 32)).to(dev)
    for prim in Hh.PRIMITIVES:
        if prim.startswith("SPA"):
            continue
        def f(prim=prim):
            op = MixedOp(32, prim).eval()
            S.load_formula_weights(op, salt=Hh.PRIMITIVES.index(prim) + 1)
            with torch.no_grad():
                y = op.to(dev)(x)
            return "maxabs %.3e (scale %.2f)" % (maxabs(y.cpu(), g[prim + ".y"]), np.abs(g[prim + ".y"]).max())
        run("prim " + prim, f)


def gf():
    g = G("gb_guided_filter")
    y = ops.to_nhwc(t(S.make_smooth_feature(21, 1, 32, 24, 32)).to(dev))
    guide = ops.channel_residue(y)
    lf = ops.guided_filter_pair(guide, y)
    out = []
    for i, eps in enumerate((1e-3, 1e-4)):
        m = lf[i].permute(0, 3, 1, 2).cpu()
        out.append("eps%g: vs32 %.2e vs64 %.2e (ref floor %.2e)" % (eps, maxabs(m, g["lf_eps%g" % eps]), maxabs(m, g["lf64_eps%g" % eps]),
                                                                   maxabs(g["lf_eps%g" % eps], g["lf64_eps%g" % eps])))
    return "; ".join(out)


def fusion():
    from oracle.paif_oracle import FUSION_AT
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    S.load_formula_weights(net)
    net = net.to(dev)
    g = G("gc_fusion_48x64")
    ir, vis, _ = S.make_batch(1, 48, 64)
    ycc = ops.rgb2ycrcb(t(vis).to(dev))
    inter = {}
    with torch.no_grad():
        fused = net(t(ir).to(dev), ycc[:, 0:1], inter=inter)
    out = ["fused %.2e" % maxabs(fused.cpu(), g["fused"])]
    for k in ("fir", "ir_feature", "vis_feature", "feature2"):
        out.append("%s %.2e" % (k, maxabs(inter[k].permute(0, 3, 1, 2).cpu(), g[k])))
    return " ".join(out)


def full():
    from oracle.paif_oracle import FUSION_AT
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    S.load_formula_weights(net)
    net = net.to(dev)
    g = G("gf_model_b3_1x480x640")
    ir, vis, _ = S.make_batch(1, 480, 640)
    ycc = ops.rgb2ycrcb(t(vis).to(dev))
    with torch.no_grad():
        fused = net(t(ir).to(dev), ycc)
    r = "fused480x640 %.2e" % maxabs(fused.cpu(), g["fused"])
    # quick timing at B=8
    irb = torch.rand(8, 1, 480, 640, device=dev)
    yb = torch.rand(8, 1, 480, 640, device=dev)
    with torch.no_grad():
        for _ in range(2):
            net(irb, yb)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            net(irb, yb)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    return r + "  | B=8 fwd %.2f ms -> %.1f pairs/s" % (dt * 1e3, 8 / dt)


def full64():
    from oracle.paif_oracle import FUSION_AT
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
    sd = {k: t(S.formula_tensor("enhance_net." + k, tuple(v.shape))).to(v.dtype) for k, v in net.state_dict().items()}
    net.load_state_dict(sd); net = net.to(dev)
    g = G("gf_model_b3_1x480x640")
    ir, vis, _ = S.make_batch(1, 480, 640)
    ycc = ops.rgb2ycrcb(t(vis).to(dev))
    with torch.no_grad():
        fused = net(t(ir).to(dev), ycc)
    return "fused vs ref32 %.2e vs ref64 %.2e (ref floor %.2e)" % (maxabs(fused.cpu(), g["fused"]), maxabs(fused.cpu(), g["fused64"]), maxabs(g["fused"], g["fused64"]))


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "lib", ops.lib().paif_version(), "CUs", ops.lib().paif_device_cus())
    for prec in ("f32", "bf16x3"):
        ops.set_conv_precision(prec)
        print("=== conv precision", prec, flush=True)
        prims()
        run("guided filter", gf)
        run("fusion 48x64", fusion)
        run("fusion 480x640 weights", full64)
        run("fusion 480x640 + timing", full)
