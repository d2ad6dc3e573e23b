"""Composite forward at batch size B (default 1) x N, for rocprofv3: fwd_small_batch.py [B] [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import synthetic as S
from paif_amd.core.model_fusion_auto import Network_MM_Searched
from oracle.paif_oracle import FUSION_AT   # genotype constant only
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
net = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
S.load_formula_weights(net)
net = net.to(dev)
ir, vis, _ = S.make_batch(B, 480, 640)
ir, vis = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)
with torch.no_grad():
    for _ in range(N):
        net(ir, vis)
torch.cuda.synchronize()
