"""Time the one-kernel ResidualDenseBlock (csrc/rdb_fused.hip) against the three-launch form inside an fp16 forward-like loop.
    python tools/rdb_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S  # noqa: E402
from paif_amd.operations_m import ResidualDenseBlock  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
g = torch.Generator().manual_seed(1)
m = ResidualDenseBlock(32, 3, 1).eval()
S.load_formula_weights(m)
m = m.to(dev)
x = ops.cast_storage(torch.from_numpy(S.make_smooth_feature(3, B, 32, H, W)).permute(0, 2, 3, 1).contiguous().to(dev), torch.float16)
res = (x.clone(), x.clone())
ops.set_storage("f16")
for nres in (0, 2):
    for fused in (False, True):
        ops.CONFIG["rdb_fused"] = fused
        with ops.bf16_activations(), torch.no_grad():
            for _ in range(5):
                m.forward_nhwc(x, res[:nres], None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                m.forward_nhwc(x, res[:nres], None)
            torch.cuda.synchronize()
        print("nres %d  %s  %.1f us" % (nres, "one kernel " if fused else "three convs", (time.perf_counter() - t0) / 30 * 1e6), flush=True)
