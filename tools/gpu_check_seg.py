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


def wetr_bwd(bb):
    m = Network_MM_Searched(32, O.FUSION_AT, None, None, bb, num_classes=9).eval(); S.load_formula_weights(m); m = m.to(dev)
    g = G("ge_wetr_" + bb)
    x = t(G("gd_colour_glue")["seg_in_b2"]).to(dev).requires_grad_(True)
    logits = m.denoise_net(x)
    r = t(S.make_feature(41, tuple(logits.shape))).to(dev)
    (logits * r).sum().backward()
    d = (x.grad.cpu() - t(g["dx"])).abs()
    return "logits %.1e dx maxabs %.2e (scale %.2e) rel-l2 %.2e" % (maxabs(logits.detach().cpu(), g["logits"]), float(d.max()), float(np.abs(g["dx"]).max()),
        float(torch.linalg.norm(x.grad.cpu() - t(g["dx"])) / torch.linalg.norm(t(g["dx"]))))


def unit_bwd():
    out = []
    # layernorm bwd
    x = torch.randn(50, 320, requires_grad=True); g_ = torch.randn(320); be = torch.randn(320); dy = torch.randn(50, 320); add = torch.randn(50, 320)
    torch.nn.functional.layer_norm(x, (320,), g_, be, 1e-6).backward(dy)
    got = ops.layernorm_bwd(x.detach().to(dev), g_.to(dev), dy.to(dev), 1e-6, add=add.to(dev)).cpu()
    out.append("ln_bwd %.1e" % maxabs(got, x.grad + add))
    # dwconv gelu bwd
    x = torch.randn(2, 9, 11, 64, requires_grad=True); w = torch.randn(64, 1, 3, 3); b = torch.randn(64); dy = torch.randn(2, 9, 11, 64)
    y = torch.nn.functional.gelu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, 1, 1, 1, 64)).permute(0, 2, 3, 1)
    y.backward(dy)
    got = ops.dwconv3_bias_gelu_bwd(x.detach().to(dev), w.to(dev), b.to(dev), dy.to(dev)).cpu()
    out.append("dwgelu_bwd %.1e" % maxabs(got, x.grad))
    # attention bwd
    for (B, N, Nk, C, heads) in [(2, 200, 45, 128, 2), (1, 130, 300, 64, 1), (2, 96, 6, 64, 2)]:
        q = torch.randn(B, N, C, requires_grad=True); kv = torch.randn(B, Nk, 2 * C, requires_grad=True); do = torch.randn(B, N, C)
        hd = C // heads
        qq = q.view(B, N, heads, hd).permute(0, 2, 1, 3)
        kk = kv.view(B, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
        ref = (((qq @ kk[0].transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ kk[1]).transpose(1, 2).reshape(B, N, C)
        ref.backward(do)
        o, lse = ops.sr_attention(q.detach().to(dev), kv.detach().to(dev), heads, want_lse=True)
        dq, dkv = ops.sr_attention_bwd(q.detach().to(dev), kv.detach().to(dev), o, do.to(dev), lse, heads)
        out.append("attn_bwd(%d,%d,%d,%d) dq %.1e dkv %.1e" % (N, Nk, C, heads, maxabs(dq.cpu(), q.grad), maxabs(dkv.cpu(), kv.grad)))
    # im2col / col2im adjoint, resize adjoint
    x = torch.randn(2, 13, 17, 8, requires_grad=True)
    col = torch.nn.functional.unfold(x.permute(0, 3, 1, 2), 3, 1, 1, 2)  # [B, C*9, L]
    dcol_ref = torch.randn_like(col)
    col.backward(dcol_ref)
    # our column order is (ky,kx,c); unfold's is (c,ky,kx)
    OH, OW = ops.conv_out_size(13, 3, 2, 1), ops.conv_out_size(17, 3, 2, 1)
    dc = dcol_ref.view(2, 8, 9, OH * OW).permute(0, 3, 2, 1).reshape(2, OH, OW, 72).contiguous()
    got = ops.col2im(dc.to(dev), 2, 13, 17, 8, 3, 2, 1).cpu()
    out.append("col2im %.1e" % maxabs(got, x.grad))
    x = torch.randn(2, 5, 7, 8, requires_grad=True)
    up = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(20, 28), mode="bilinear", align_corners=False)
    dd = torch.randn(2, 20, 28, 16)
    up.backward(dd[..., 8:].permute(0, 3, 1, 2))
    got = ops.resize_bilinear_adjoint(dd.to(dev), 8, 8, 5, 7).cpu()
    out.append("resize_adj %.1e" % maxabs(got, x.grad))
    # upsample + CE
    lg = torch.randn(2, 6, 8, 9, requires_grad=True); lab = torch.randint(0, 9, (2, 24, 32)); lab[0, :3] = 255
    up = torch.nn.functional.interpolate(lg.permute(0, 3, 1, 2), size=(24, 32), mode="bilinear", align_corners=False)
    loss = torch.nn.functional.cross_entropy(up, lab, ignore_index=255)
    loss.backward()
    lc = ops.upsample_ce_fwd(lg.detach().to(dev), lab.to(dev))
    gs = (1.0 / lc[1:2]).contiguous()
    dl = ops.upsample_ce_bwd(lg.detach().to(dev), lab.to(dev), gs, cp=32).cpu()
    out.append("ce loss %.1e dlogits %.1e pad %.1e" % (abs(float(lc[0]) - float(loss)), maxabs(dl[..., :9], lg.grad), float(dl[..., 9:].abs().max())))
    return " ".join(out)


if __name__ == "__main__":
    run("unit_bwd", unit_bwd)
    run("wetr_bwd mit_b0", lambda: wetr_bwd("mit_b0"))
    run("wetr_bwd mit_b3", lambda: wetr_bwd("mit_b3"))

    for prec in ("f32", "bf16x3"):
        ops.set_gemm_precision(prec)
        print("=== gemm precision", prec, flush=True)
        run("unit", unit)
        holder = {}
        def b3():
            r, m = wetr("mit_b3"); holder["m"] = m; return r
        run("wetr mit_b3", b3)
        run("wetr_bwd mit_b3", lambda: wetr_bwd("mit_b3"))
        if "m" in holder:
            run("full 480x640", lambda: full(holder["m"]))
