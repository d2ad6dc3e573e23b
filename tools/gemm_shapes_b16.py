"""fp32 GEMM at the mit_b3 shapes of a B=16 480x640 forward (BASELINE configs[2]): time, TFLOP/s, algorithmic TB/s per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    ops.set_gemm_precision(sys.argv[2])
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0
shapes = []
for tok, D, depth, sr in ((19200, 64, 3, 8), (4800, 128, 4, 4), (1200, 320, 18, 2), (300, 512, 3, 1)):
    M = B * tok
    shapes += [("q/proj s%d" % D, M, D, D, 2 * depth), ("fc1 s%d" % D, M, 4 * D, D, depth), ("fc2 s%d" % D, M, D, 4 * D, depth),
               ("kv s%d" % D, B * 300, 2 * D, D, depth)]
    if sr > 1:
        shapes.append(("sr s%d" % D, B * 300, D, D * sr * sr, depth))
shapes += [("head fold c1", B * 19200, 256, 64, 1), ("head pred", B * 19200, 9, 256, 1), ("patch1", B * 19200, 64, 160, 1)]
for name, M, N, K, cnt in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    t = timeit(lambda: ops.gemm(a, w, shift=bias))
    t2 = timeit(lambda: torch.addmm(bias, a, w.t()))
    by = 4 * (M * K + N * K + M * N)
    tot += t * cnt
    print("%-14s M=%7d N=%5d K=%5d x%2d  paif %8.1f us  %6.1f TF  %5.2f TB/s | hipBLASLt %8.1f us" % (name, M, N, K, cnt, t, 2 * M * N * K / t / 1e6, by / t / 1e6, t2))
print("sum over the forward: %.2f ms" % (tot / 1e3))
