"""hipGraph capture of the composite forward at small batch (the reference harness evaluates with batch_size 1):
eager vs graph-replay timing and bit-equality.  Usage: gpu_graph_check.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import synthetic as S
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from paif_amd.graph import GraphedForward
from oracle.paif_oracle import FUSION_AT   # genotype constant only (test infrastructure)

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
S.load_formula_weights(net)
net = net.to(dev)
ir, vis, _ = S.make_batch(B, 480, 640)
ir, vis = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

with torch.no_grad():
    t_eager, (f0, s0) = timeit(lambda: net(ir, vis))
g = GraphedForward(net, ir, vis)
t_graph, (f1, s1) = timeit(lambda: g(ir, vis))
print("B=%d  eager %.3f ms/forward   graph replay %.3f ms/forward   speed-up %.2fx" % (B, t_eager, t_graph, t_eager / t_graph))
print("fused equal:", torch.equal(f0, f1), " logits equal:", torch.equal(s0, s1))
# new inputs through the same graph
ir2, vis2, _ = S.make_batch(B, 480, 640, seed0=7) if "seed0" in S.make_batch.__code__.co_varnames else (None, None, None)
if ir2 is not None:
    ir2, vis2 = torch.from_numpy(ir2).to(dev), torch.from_numpy(vis2).to(dev)
    with torch.no_grad():
        fe, se = net(ir2, vis2)
    fg, sg = g(ir2, vis2)
    print("second input: fused equal:", torch.equal(fe, fg), " logits equal:", torch.equal(se, sg))
