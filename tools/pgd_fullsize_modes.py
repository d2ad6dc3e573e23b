import sys, os, torch
sys.path.insert(0, os.getcwd())
from tests.test_benchmarked_size_gpu import _model, _dev, EPS, ALPHA, S, t
from paif_amd import ops
from paif_amd.attack.attack import attack_both
dev=_dev(); m=_model()
ir,vis,lab=S.make_batch(2,480,640)
kw=dict(epsilon=EPS, alpha=ALPHA, attack_iters=5, attack_loss="l_seg", attack_way="PGD", delta0_ir=t(S.make_delta0(7, ir.shape, EPS)).to(dev), delta0_vis=t(S.make_delta0(107, vis.shape, EPS)).to(dev))
tr={}
for mode,cfg in (("f16x3",{}),("x6",dict(attack_fwd_f16x3=False,attack_bwd_f16x3=False,attn_f16x3=False)),("exact",{})):
    old=dict(ops.CONFIG); ops.CONFIG.update(cfg); ops.set_attack_precision("exact" if mode=="exact" else "bf16x6")
    tr[mode]=[]
    with torch.no_grad(): attack_both(m,t(vis).to(dev),t(ir).to(dev),t(lab).to(dev),trace=tr[mode],**kw)
    ops.CONFIG.clear(); ops.CONFIG.update(old)
for mode in ("f16x3","x6"):
    print(mode, [ (round(float((torch.sign(a["g_ir"])!=torch.sign(b["g_ir"])).float().mean()),6), round(float((torch.sign(a["g_vis"])!=torch.sign(b["g_vis"])).float().mean()),6), "%.1e"%(abs(a["loss"]-b["loss"])/abs(b["loss"]))) for a,b in zip(tr[mode],tr["exact"])])
