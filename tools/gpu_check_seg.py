"""Debug harness (GPU box): SegFormer parity numbers + timing, never stops early."""
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
from paif_amd.core.model_fusion_auto import Network_MM_Searched

def run(name, fn):
    try:
        t0 = time.time(); r = fn(); torch.cuda.synchronize()
        print("%-28s %s (%.2fs)" % (name, r, time.time() - t0), flush=True)
    except Exception:
        print("%-28s EXC\n%s" % (name, traceback.format_exc()), flush=True)

def unit():
    # GEMM / LN / attention unit checks vs torch CPU
    out = []
    a = torch.randn(300, 96); w = torch.randn(70, 96); b = torch.randn(70); r = torch.randn(300, 70)
    y = ops.gemm(a.to(dev), w.to(dev), shift=b.to(dev), res=r.to(dev)).cpu()
    out.append("gemm %.1e" % maxabs(y, a @ w.t() + b + r))
    y = ops.gemm(a.to(dev), w.to(dev), shift=b.to(dev), act=ops.ACT_GELU if hasattr(ops, 'ACT_GELU') else 1).cpu()
    out.append("gemm+gelu %.1e" % maxabs(y, torch.nn.functional.gelu(a @ w.t() + b)))
    x = torch.randn(77, 320); g_ = torch.randn(320); be = torch.randn(320)
    y = ops.layernorm(x.to(dev), g_.to(dev), be.to(dev), 1e-6).cpu()
    out.append("ln %.1e" % maxabs(y, torch.nn.functional.layer_norm(x, (320,), g_, be, 1e-6)))
    for (B, N, Nk, C, heads) in [(2, 200, 45, 128, 2), (1, 130, 300, 64, 1), (2, 96, 6, 64, 2)]:
        q = torch.randn(B, N, C); kv = torch.randn(B, Nk, 2 * C)
        o = ops.sr_attention(q.to(dev), kv.to(dev), heads).cpu()
        hd = C // heads
        qq = q.view(B, N, heads, hd).permute(0, 2, 1, 3)
        kk = kv.view(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
        ref = ((qq @ kk[0].transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ kk[1]
        ref = ref.transpose(1, 2).reshape(B, N, C)
        out.append("attn(%d,%d,%d,%d) %.1e" % (N, Nk, C, heads, maxabs(o, ref)))
    x = torch.randn(2, 13, 17, 64); w = torch.randn(64, 1, 3, 3); b = torch.randn(64)
    y = ops.dwconv3_bias_gelu(x.to(dev), w.to(dev), b.to(dev)).cpu()
    ref = torch.nn.functional.gelu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, 1, 1, 1, 64)).permute(0, 2, 3, 1)
    out.append("dwgelu %.1e" % maxabs(y, ref))
    x = torch.randn(2, 5, 7, 8); dst = torch.zeros(2, 20, 28, 16, device=dev)
    ops.resize_bilinear_into(x.to(dev), dst, 8)
    ref = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(20, 28), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    out.append("resize %.1e" % maxabs(dst[..., 8:].cpu(), ref))
    return " ".join(out)

def wetr(bb):
    m = Network_MM_Searched(32, O.FUSION_AT, None, None, bb, num_classes=9).eval(); S.load_formula_weights(m); m = m.to(dev)
    g = G("ge_wetr_" + bb)
    x = t(G("gd_colour_glue")["seg_in_b2"]).to(dev)
    with torch.no_grad():
        feats = m.denoise_net.encoder(x); logits = m.denoise_net.decoder(feats)
    r = " ".join("%s %.1e" % (k, maxabs(f.cpu(), g[k])) for k, f in zip(("c1", "c2", "c3", "c4"), feats))
    return r + " logits %.1e" % maxabs(logits.cpu(), g["logits"]), m

def full(m):
    g = G("gf_model_b3_1x480x640")
    ir, vis, _ = S.make_batch(1, 480, 640)
    with torch.no_grad():
        fused, seg = m(t(ir).to(dev), t(vis).to(dev))
    r = "fused32 %.1e fused64 %.1e logits32 %.1e logits64 %.1e (floor %.1e)" % (
        maxabs(fused.cpu(), g["fused"]), maxabs(fused.cpu(), g["fused64"]), maxabs(seg.cpu(), g["logits"]), maxabs(seg.cpu(), g["logits64"]),
        maxabs(g["logits"], g["logits64"]))
    irb = torch.rand(8, 1, 480, 640, device=dev); vb = torch.rand(8, 3, 480, 640, device=dev)
    with torch.no_grad():
        for _ in range(2): m(irb, vb)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): m(irb, vb)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    return r + " | B=8 fusion+seg fwd %.1f ms -> %.1f pairs/s" % (dt * 1e3, 8 / dt)

if __name__ == "__main__":
    run("unit", unit)
    run("wetr mit_b0", lambda: wetr("mit_b0")[0])
    holder = {}
    def b3():
        r, m = wetr("mit_b3"); holder["m"] = m; return r
    run("wetr mit_b3", b3)
    if "m" in holder:
        run("full 480x640", lambda: full(holder["m"]))
