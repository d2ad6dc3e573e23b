"""GEMMs at the mit_b3 shapes of a B=16 480x640 forward (BASELINE configs[2]): time, TFLOP/s, algorithmic TB/s per shape, for the
tile forms of the split-bf16 kernel (gemm_mfma.hip 128x64 / gemm_split2.hip 128 x 64*nt) and hipBLASLt fp32.
  python tools/gemm_shapes_b16.py [B] [precision: auto | bf16x3 | bf16x6 | auto6 | f32] [nt to force: 1 2 4 5]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    ops.set_gemm_precision(sys.argv[2])
force = int(sys.argv[3]) if len(sys.argv) > 3 else True
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = tot2 = 0.0
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
    ops.CONFIG["gemm2"] = False
    y1 = ops.gemm(a, w, shift=bias, res=res, act=ops.ACT_GELU)
    t = timeit(lambda: ops.gemm(a, w, shift=bias))
    ops.CONFIG["gemm2"] = force
    y2 = ops.gemm(a, w, shift=bias, res=res, act=ops.ACT_GELU)
    t3 = timeit(lambda: ops.gemm(a, w, shift=bias))
    same = bool(torch.equal(y1, y2))
    t2 = timeit(lambda: torch.addmm(bias, a, w.t()))
    by = 4 * (M * K + N * K + M * N)
    tot += t * cnt
    tot2 += t3 * cnt
    print("%-14s M=%7d N=%5d K=%5d x%2d  128x64 %7.1f us %6.1f TF %5.2f TB/s | wide %7.1f us %6.1f TF %5.2f TB/s %s | hipBLASLt %7.1f us"
          % (name, M, N, K, cnt, t, 2 * M * N * K / t / 1e6, by / t / 1e6, t3, 2 * M * N * K / t3 / 1e6, by / t3 / 1e6,
             "bit-equal" if same else "DIFFERENT", t2))
print("sum over the forward: 128x64 %.2f ms, wide %.2f ms" % (tot / 1e3, tot2 / 1e3))
