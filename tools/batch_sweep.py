"""Per-pair time of the fusion forward as a function of the batch size (fp16 / fp32 storage): do maps that fit the 256 MiB Infinity
Cache (fp16 maps: 19.7 MB per image) make the per-pair time drop?   python tools/batch_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, synthetic as S  # noqa: E402
from paif_amd.core.model_fusion_auto import Network_Fusion_Searched  # noqa: E402
from paif_amd.genotypes import FUSION_AT  # noqa: E402

dev = torch.device("cuda:0")
net = Network_Fusion_Searched(32, None, FUSION_AT).eval()
S.load_formula_weights(net)
net = net.to(dev)
for storage in ("f16", "f32"):
    ops.set_storage(storage)
    for B in (1, 2, 3, 4, 6, 8, 12, 16):
        ir, vis, _ = S.make_batch(B, 480, 640)
        irt, vist = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)
        with torch.no_grad():
            for _ in range(5):
                net(irt, ops.rgb2ycrcb(vist))
            torch.cuda.synchronize()
            n = max(10, 160 // B)
            t0 = time.perf_counter()
            for _ in range(n):
                net(irt, ops.rgb2ycrcb(vist))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print("%s B=%2d  %.3f ms/step  %.4f ms/pair  %.0f pairs/s" % (storage, B, dt * 1e3, dt * 1e3 / B, B / dt), flush=True)
