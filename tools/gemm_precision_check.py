import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import synthetic as S, ops
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from oracle.paif_oracle import FUSION_AT
dev = torch.device("cuda:0")
net = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
S.load_formula_weights(net); net = net.to(dev)
for B in (1, 16):
    ir, vis, _ = S.make_batch(B, 480, 640)
    ir, vis = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)
    outs = {}
    for prec in ("f32", "bf16x3"):
        ops.set_gemm_precision(prec)
        with torch.no_grad():
            for _ in range(3): net(ir, vis)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): f, s = net(ir, vis)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
        outs[prec] = s
        print("B=%d gemm %s: %.2f ms/forward" % (B, prec, dt))
    d = (outs["f32"] - outs["bf16x3"]).abs().max().item(); sc = outs["f32"].abs().max().item()
    print("   logits max|d| %.2e (range %.2f), argmax agreement %.5f" % (d, sc, (outs["f32"].argmax(1) == outs["bf16x3"].argmax(1)).float().mean().item()))
