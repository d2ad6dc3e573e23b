"""Micro-benchmark of the fp32 GEMM at the MiT shapes of a B=1 480x640 forward: gemm_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(300, 512, 2048), (300, 2048, 512), (300, 512, 512), (1200, 320, 320), (1200, 1280, 320), (1200, 320, 1280),
                  (4800, 128, 128), (19200, 64, 64), (300, 512, 1280), (76800, 256, 1024)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    t = timeit(lambda: ops.gemm(a, w))
    ref = a @ w.t()
    err = (ops.gemm(a, w) - ref).abs().max().item() / ref.abs().max().item()
    t2 = timeit(lambda: torch.matmul(a, w.t()))
    print("M=%6d N=%5d K=%5d  paif %7.1f us (%.1f TF)   torch/hipBLASLt %7.1f us   rel err %.1e" % (M, N, K, t, 2 * M * N * K / t / 1e6, t2, err))
