cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, subprocess, sys, os, time
import torch
sys.path.insert(0, ".")
from paif_amd import ops, synthetic as S
from paif_amd.core.model_fusion_auto import Network_Fusion_Searched
from paif_amd.genotypes import FUSION_AT
dev = torch.device("cuda:0")
net = Network_Fusion_Searched(32, None, FUSION_AT).eval(); S.load_formula_weights(net); net = net.to(dev)
ir, vis, _ = S.make_batch(8, 480, 640)
irt, vist = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)
ops.set_storage("f16")
def run(n=40):
    with torch.no_grad():
        for _ in range(8): net(irt, ops.rgb2ycrcb(vist))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): net(irt, ops.rgb2ycrcb(vist))
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for fused in (False, True):
        ops.CONFIG["rdb_fused"] = fused
        print("rdb_fused", fused, "%.4f ms/step" % run(), flush=True)
PY
