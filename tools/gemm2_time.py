"""Time csrc/gemm_split2.hip on the stage-3 / stage-4 shapes of a B=16 mit_b3 forward (one line per shape; PAIF_LIB selects a variant build)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_gemm_precision(prec)
force = int(sys.argv[2]) if len(sys.argv) > 2 else True
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for name, M, N, K in (("qproj320", 19200, 320, 320), ("fc1_320", 19200, 1280, 320), ("fc2_320", 19200, 320, 1280), ("fc1_512", 4800, 2048, 512),
                      ("fc2_128", 76800, 128, 512), ("fc1_128", 76800, 512, 128), ("big", 153600, 320, 320)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    ops.CONFIG["gemm2"] = force
    t = timeit(lambda: ops.gemm(a, w, shift=bias))
    out.append("%s %.1f (%.0f TF)" % (name, t, 2 * M * N * K / t / 1e6))
print(os.environ.get("PAIF_LIB", "default"), " | ".join(out))
