import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
from paif_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, sc) in ((19200, 320, 320, 1.0), (19200, 1280, 320, 1.0), (19200, 320, 1280, 1.0), (4096, 64, 4096, 0.01), (4096, 256, 512, 100.0)):
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * sc).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    ref = a.double() @ w.double().t()
    out = {}
    for prec in ("f32", "bf16x3", "bf16x6", "f16x3"):
        ops.set_gemm_precision(prec)
        y = ops.gemm(a, w)
        e = (y.double() - ref).abs()
        out[prec] = (float(e.max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    print(M, N, K, sc, {k: "%.2e / %.2e" % v for k, v in out.items()})
