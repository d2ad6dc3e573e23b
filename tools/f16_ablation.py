"""What each design choice of the fp16 storage mode buys, on the REAL kernels: argmax agreement with the reference on the eight
480x640 samples of the benchmarked batch (tests/golden/gp_model_b3_8x480x640.npz), logit error on sample 0.

    python tools/f16_ablation.py [--out gpurun_out/f16_ablation.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from paif_amd import ops, synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import helpers as Hh
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT
    from paif_amd.util.util import ConfusionMeter

    g = dict(np.load(os.path.join(Hh.GOLDEN, "gf_model_b3_1x480x640.npz")))
    g8 = dict(np.load(os.path.join(Hh.GOLDEN, "gp_model_b3_8x480x640.npz")))
    dev = torch.device("cuda:0")
    t = Hh.t
    ir, vis, lab = S.make_batch(8, 480, 640)
    rng = float(g["logits"].max() - g["logits"].min())
    table = {}

    def run(name, storage, **cfg):
        old = {k: ops.CONFIG.get(k) for k in cfg}
        ops.CONFIG.update(cfg)
        ops.set_storage(storage)
        m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
        S.load_formula_weights(m, head=Hh.HEAD480)
        m = m.to(dev)
        moved = []
        for i in range(8):
            with torch.no_grad():
                fused, seg = m(t(ir[i:i + 1]).to(dev), t(vis[i:i + 1]).to(dev))
            pred = ConfusionMeter(9, dev).update(seg, t(lab[i:i + 1]).to(dev)).cpu().numpy()
            moved.append(int((pred[0] != g8["pred"][i]).sum()))
            if i == 0:
                lerr = float((seg.cpu() - t(g["logits"])).abs().mean()) / rng
                ferr = float((fused.cpu().double() - t(g["fused64"]).double()).abs().mean())
        for k, v in old.items():
            if v is None:
                ops.CONFIG.pop(k, None)
            else:
                ops.CONFIG[k] = v
        ops.set_storage("f32")
        table[name] = dict(moved=moved, agreement_8=1.0 - sum(moved) / (8 * 307200.0), agreement_min=1.0 - max(moved) / 307200.0,
                           logits_mean_over_range_sample0=lerr, fused_mean_abs_sample0=ferr)
        print("%-58s moved %-52s 8 samples %.5f  min %.5f  logits mean %.2e  fused mean %.2e" % (
            name, moved, table[name]["agreement_8"], table[name]["agreement_min"], lerr, ferr), flush=True)

    run("fp32 storage", "f32")
    run("fp16 storage (as shipped)", "f16")
    run("fp16, last map stored as fp16 too", "f16", f16_last_f32=False)
    run("fp16, folded 1x1 with plain fp16 weights (hi pieces only)", "f16", f16_decomp_split=False)
    run("fp16, both of the above", "f16", f16_last_f32=False, f16_decomp_split=False)
    run("bf16 storage (maps + weights)", "bf16")
    run("bf16 maps, split-bf16 weights", "bf16_split")
    if args.out:
        json.dump(table, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
