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

# the 32-channel convs: x6 / f16x3 / x3 / exact against float64 (torch conv on the host)
import torch.nn.functional as F
for (kh, dil, nsrc, sc) in ((3, 1, 1, 1.0), (3, 1, 3, 1.0), (7, 1, 1, 1.0), (1, 1, 3, 1.0), (3, 2, 1, 0.02)):
    g = torch.Generator().manual_seed(kh * 10 + nsrc)
    xs = [(torch.randn(2, 96, 128, 32, generator=g) * sc).to(dev) for _ in range(nsrc)]
    w = (torch.randn(32, 32 * nsrc, kh, kh, generator=g) * 0.05).to(dev)
    pad = dil * (kh - 1) // 2
    ref = F.conv2d(torch.cat([x.cpu().double() for x in xs], dim=3).permute(0, 3, 1, 2), w.cpu().double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
    out = {}
    for prec in ("f32", "bf16x3", "bf16x6", "f16x3"):
        wp = ops.pack_conv_weight(w, nsrc, 32, kh, precision=prec)
        y = ops.conv2d(xs, wp, kh, dil)
        e = (y.cpu().double() - ref).abs()
        out[prec] = (float(e.max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    print("conv k%d d%d x%d scale %g" % (kh, dil, nsrc, sc), {k: "%.2e / %.2e" % v for k, v in out.items()})
