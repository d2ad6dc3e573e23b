"""Debug harness (GPU box): input-gradient parity of the fusion net and the primitives."""
import os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paif_amd import ops, synthetic as S
from oracle import paif_oracle as O
from tests import helpers as Hh
from tests.helpers import t, maxabs
dev = torch.device("cuda:0")
G = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))

def run(name, fn):
    try:
        t0 = time.time(); r = fn(); torch.cuda.synchronize()
        print("%-30s %s (%.2fs)" % (name, r, time.time() - t0), flush=True)
    except Exception:
        print("%-30s EXC\n%s" % (name, traceback.format_exc()), flush=True)

def prims():
    from paif_amd.core.model_fusion_auto import MixedOp
    g = G("ga_primitives")
    for prim in ("Denseblocks_3_1", "DilConv_3_2", "ECAattention_3", "Residualblocks_7_1", "DilConv_5_1", "Denseblocks_5_2",
                 "Denseblocks_7_1", "Residualblocks_3_2", "Residualblocks_5_2"):
        def f(prim=prim):
            op = MixedOp(32, prim).eval(); S.load_formula_weights(op, salt=Hh.PRIMITIVES.index(prim) + 1); op.to(dev)
            x = t(S.make_smooth_feature(11, 1, 32, 24, 32)).to(dev).requires_grad_(True)
            y = op(x)
            (y * t(S.make_feature(12, (1, 32, 24, 32))).to(dev)).sum().backward()
            return "y %.1e dx %.2e (scale %.2f)" % (maxabs(y.detach().cpu(), g[prim + ".y"]), maxabs(x.grad.cpu(), g[prim + ".dx"]), np.abs(g[prim + ".dx"]).max())
        run("prim bwd " + prim, f)

def fusion():
    from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
    g = G("gc_fusion_2x64x96")
    net = Network_Fusion_Searched(32, None, O.FUSION_AT).eval(); S.load_formula_weights(net)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(dev)
    ir, vis, _ = S.make_batch(2, 64, 96)
    ycc = O.rgb2ycrcb(t(vis))
    irt = t(ir).to(dev).requires_grad_(True); yt = ycc[:, 0:1].contiguous().to(dev).requires_grad_(True)
    fused = net(irt, yt)
    r = t(S.make_feature(31, tuple(fused.shape)))
    (fused * r.to(dev)).sum().backward()
    # fp64 oracle for the noise floor
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    i64 = t(ir).double().requires_grad_(True); y64 = ycc[:, 0:1].double().clone().requires_grad_(True)
    f64 = O.fusion_forward(i64, y64, sd64)
    (f64 * r.double()).sum().backward()
    out = ["fused %.1e" % maxabs(fused.detach().cpu(), g["fused"])]
    for name, mine, ref32, ref64 in (("d_ir", irt.grad, g["d_ir"], i64.grad), ("d_y", yt.grad, g["d_y"], y64.grad)):
        out.append("%s: vs ref32 %.2e | vs fp64 %.2e (ref32 vs fp64 floor %.2e, scale %.2f)" % (
            name, maxabs(mine.cpu(), ref32), maxabs(mine.cpu().double(), ref64), maxabs(t(ref32).double(), ref64), np.abs(ref32).max()))
    return " ".join(out)

if __name__ == "__main__":
    for prec in ("f32", "bf16x3"):
        ops.set_conv_precision(prec)
        print("=== conv precision", prec, flush=True)
        prims()
        run("fusion bwd 2x64x96", fusion)
