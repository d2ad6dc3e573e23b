"""Child process of tests/test_00_rccl_gpu.py: a 1-rank `nccl` (= RCCL) process group on the MI355X, brought up BEFORE any other
GPU call of this process, then the adversarial-training step's gradient path of BASELINE configs[4] with and without the bucketed
all-reduce (paif_amd.dist_utils.GradAllReduce: side stream, event hand-off, ReduceOp.AVG, work.wait()).  Prints ONE JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # before any other GPU call
    torch.cuda.set_device(0)

    import numpy as np
    from paif_amd import ops, synthetic as S
    from paif_amd.attack.attack import attack_both
    from paif_amd.core.loss import Fusionloss_grad2
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.dist_utils import GradAllReduce
    from paif_amd.genotypes import FUSION_AT
    from paif_amd.utils.optimizer import PolyWarmupAdamW

    net = Network_MM_Searched(32, FUSION_AT, Fusionloss_grad2(), torch.nn.CrossEntropyLoss(ignore_index=255), "mit_b0", num_classes=9)
    S.load_formula_weights(net)
    net = net.to(dev)
    ir_np, vis_np, lab_np = S.make_batch(2, 64, 96)
    ir, vis, lab = (torch.from_numpy(a).to(dev) for a in (ir_np, vis_np, lab_np))
    mask = torch.from_numpy(np.maximum(ir_np, vis_np[:, :1]).astype("float32")).to(dev)
    pg = net.denoise_net.get_param_groups()
    opt = PolyWarmupAdamW(params=[dict(params=pg[0], lr=8e-5, weight_decay=0.01), dict(params=pg[1], lr=8e-5, weight_decay=0.0),
                                  dict(params=pg[2], lr=8e-4, weight_decay=0.01),
                                  dict(params=list(net.enhance_net.parameters()), lr=8e-5, weight_decay=0.01)],
                          lr=8e-5, weight_decay=0.01, betas=(0.9, 0.999), warmup_iter=3000, max_iter=160000, warmup_ratio=1e-5, power=1.0)
    arena = opt.arena()
    d0i = torch.from_numpy(S.make_delta0(0, ir_np.shape, 8 / 255.)).to(dev)
    d0v = torch.from_numpy(S.make_delta0(100, vis_np.shape, 8 / 255.)).to(dev)

    def attack():
        net.eval()
        with torch.no_grad():
            return attack_both(net, vis, ir, lab, attack_loss="l_seg", attack_iters=2, epsilon=8 / 255., alpha=2 / 255., attack_way="PGD",
                               delta0_ir=d0i, delta0_vis=d0v)

    def backward(reducer, d_ir, d_vis):
        net.train()
        ops.DROP_RNG.reseed(7, rank=0, step=0)
        opt.zero_grad()
        loss = net._loss_coupled((ops.add(ir, d_ir.detach()), ir), (ops.add(vis, d_vis.detach()), vis), mask, lab)
        if reducer is not None:
            reducer.begin()
        loss.backward()
        early = len(reducer.launched) if reducer is not None else 0
        launched = list(reducer.launched) if reducer is not None else []
        if reducer is not None:
            launched_all = None
            # finish() resets `launched`: read the full list through a wrapper
            orig = reducer._launch
            seen = list(reducer.launched)

            def spy(b, _orig=orig, _seen=seen):
                _orig(b)
                _seen.append(reducer.buckets[b])
            reducer._launch = spy
            reducer.finish()
            reducer._launch = orig
            launched_all = seen
        else:
            launched_all = []
        torch.cuda.synchronize()
        return float(loss), arena.grad.clone(), early, launched_all

    # pass A: plain backward, no reducer anywhere
    d_ir, d_vis = attack()
    loss_a, grad_a, _, _ = backward(None, d_ir, d_vis)
    # pass B: reducer installed for the WHOLE step -- the attack's input-gradient reverse passes run with the hook in place
    # (they must not mark a single milestone), then the armed training backward
    red = GradAllReduce(arena, model=net, bucket_mb=1.0).install()
    marks = []
    inner = red.mark_ready

    def counting(module, _inner=inner):
        marks.append((type(module).__name__, red.armed))
        _inner(module)
    ops.GRAD_READY[0] = counting
    d_ir2, d_vis2 = attack()
    marks_during_attack = len(marks)
    launched_during_attack = len(red.launched)
    loss_b, grad_b, early, launched = backward(red, d_ir2, d_vis2)
    ops.GRAD_READY[0] = None
    milestones = len(red.module_params)
    # global_minmax=True (SURVEY 8(e) optional mode): the glue's batch-global min/max goes through a 2-float all-reduce(MIN) on the
    # device; over a 1-rank group it must reproduce the plain forward bit for bit (and the oracle's batch-global semantics, which
    # tests/test_seg_gpu.py pins for the plain forward)
    net.eval()
    with torch.no_grad():
        f_plain, s_plain = net(ir, vis)
        net.global_minmax = True
        f_glob, s_glob = net(ir, vis)
        net.global_minmax = False
    gm_equal = bool(torch.equal(f_plain, f_glob) and torch.equal(s_plain, s_glob))
    out = {
        "global_minmax_equals_plain_forward": gm_equal,
        "backend": dist.get_backend(), "world": dist.get_world_size(), "device": torch.cuda.get_device_name(0),
        "buckets": len(red.buckets), "bucket_floats": [e - s for s, e in red.buckets],
        "marks_during_attack": marks_during_attack, "launched_during_attack": launched_during_attack,
        "marks_during_backward": len(marks) - marks_during_attack, "milestones": milestones,
        "launched_before_backward_returned": early, "launched_total": len(launched),
        "launched_each_bucket_once": sorted(launched) == sorted(red.buckets) and len(set(launched)) == len(launched),
        "launch_order_is_arena_order": launched == sorted(launched),
        "loss_a": loss_a, "loss_b": loss_b,
        "grad_max_abs_diff": float((grad_a - grad_b).abs().max()), "grad_abs_max": float(grad_a.abs().max()),
        "delta_equal": bool(torch.equal(d_ir, d_ir2) and torch.equal(d_vis, d_vis2)),
        "live_floats": red.live_end, "arena_floats": int(arena.grad.numel()),
    }
    print("NCCL_CHILD " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
