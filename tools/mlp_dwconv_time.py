"""HIP-event time of Mlp.dwconv + GELU (forward, and the input-gradient pass) at the four mit_b3 stage shapes of a B=8 480x640 pair
batch, weighted by the blocks per stage (3, 4, 18, 3): python tools/mlp_dwconv_time.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
g = torch.Generator().manual_seed(0)
tot_f = tot_b = 0.0
for (H, W, C, nblk) in [(120, 160, 256, 3), (60, 80, 512, 4), (30, 40, 1280, 18), (15, 20, 2048, 3)]:
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(dev)
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    dy = torch.randn(B, H, W, C, generator=g).to(dev)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    tf = timeit(lambda: ops.dwconv3_bias_gelu(x, w, bias))
    tb = timeit(lambda: ops.dwconv3_bias_gelu_bwd(x, w, bias, dy))
    mb = B * H * W * C * 4 / 1e6
    print("%3dx%3dx%4d: fwd %6.1f us (%4.2f TB/s over 2 maps) | bwd %6.1f us (%4.2f TB/s over 3 maps)" % (H, W, C, tf, 2 * mb / tf, tb, 3 * mb / tb))
    tot_f += nblk * tf
    tot_b += nblk * tb
print("%s per mit_b3 pass (28 blocks): fwd %.2f ms, bwd %.2f ms" % (os.path.basename(os.environ.get("PAIF_LIB", "default")), tot_f / 1e3, tot_b / 1e3))
