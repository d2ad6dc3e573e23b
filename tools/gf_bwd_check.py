"""Guided-filter reverse pass: the streaming form (gf_backward.hip) against the round-1 kernels (PAIF_GF_BWD=v1), stage by stage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paif_amd import ops, _lib
from paif_amd.ops import _p, _stream
dev = torch.device("cuda:0")
L = _lib.load()
for (B, H, W, use_add) in [(1, 20, 24, False), (2, 33, 41, True), (1, 64, 96, True), (3, 10, 10, False), (1, 70, 130, True), (2, 480, 100, False)]:
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    xn = torch.randn(B, H, W, 32, generator=g).to(dev)
    guide = ops.channel_residue(xn)
    lf, ab = ops.guided_filter_pair(guide, xn, want_ab=True, tape="ab")
    dlf = torch.randn(2, B, H, W, 32, generator=g).to(dev)
    add = torch.randn(B, H, W, 32, generator=g).to(dev) if use_add else None
    out = {}
    for mode in ("v1", "v2"):
        os.environ["PAIF_GF_BWD"] = mode
        gstat = torch.zeros(L.paif_guided_filter_fused_workspace_floats(B, H, W), device=dev)
        t_my, t_mgy, dy = torch.zeros_like(xn), torch.zeros_like(xn), torch.zeros_like(xn)
        t_g = torch.zeros(B, H, W, 4, device=dev)
        _lib.check(L.paif_guided_filter_bwd_input(_p(guide), _p(xn), _p(ab), _p(dlf), 1e-3, 1e-4, _p(add), _p(gstat), _p(t_my), _p(t_mgy),
                                                  _p(t_g), _p(dy), B, H, W, _stream()), "bwd")
        torch.cuda.synchronize()
        out[mode] = (t_my, t_mgy, t_g, dy)
    msg = []
    for name, a, b in zip(("t_my", "t_mgy", "t_g", "dy"), out["v1"], out["v2"]):
        d = (a - b).abs()
        msg.append("%s %.2e/%.2e" % (name, d.max().item(), a.abs().max().item()))
        if name == "t_g":
            msg.append("[x %.1e y %.1e z %.1e]" % tuple(d[..., i].max().item() for i in range(3)))
    print((B, H, W, use_add), " ".join(msg), flush=True)
