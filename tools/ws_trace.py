"""Timeline of the wave-specialised conv kernel (debug build with -DPAIF_WS_TRACE): ws_trace.py kh dil nsrc nres"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, _lib
kh, dil, nsrc, nres = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
B, H, W = 8, 480, 640
x = [torch.randn(B, H, W, 32, device=dev) for _ in range(nsrc)]
r = [torch.randn(B, H, W, 32, device=dev) for _ in range(nres)]
out = torch.empty(B, H, W, 32, device=dev)
w = torch.randn(32, 32 * nsrc, kh, kh, device=dev) * 0.05
wpk = ops.pack_conv_weight(w, nsrc, 32, kh)
for _ in range(3):
    ops.conv2d(x, wpk, kh, dil=dil, res=tuple(r), out=out)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 512)()
assert L.paif_debug_ws_trace(buf) == 0
t0 = buf[0]
print("stage | L:start commit_done | S:start stored | M:start mfma_issued parked   (cycles since loader start of stage 0)")
for st in range(2, 26):
    v = [int(buf[st * 8 + k]) - int(t0) for k in range(8)]
    print("%3d | %7d %7d | %7d %7d | %7d %7d %7d" % (st, v[0], v[1], v[2], v[3], v[4], v[5], v[6]))
