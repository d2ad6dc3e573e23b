"""Round 6: would the segmentation net's inference forward survive 16-bit token maps?

Emulation on the REAL kernels: the chosen classes of MiT maps are rounded to IEEE fp16 (or bf16) where they are produced / consumed
(monkeypatched ops entry points, fp32 storage otherwise), the fusion network runs in its fp16 storage mode (the configuration the
headline is measured in), and the x4-upsampled argmax is compared with the REFERENCE's predictions on the first N samples of
tests/golden/gq_model_b3_32x480x640.npz (fallback: gp_model_b3_8x480x640.npz).

    python tools/seg_storage_sensitivity.py [--samples 8] [--out gpurun_out/seg_storage_sensitivity.json]

classes:  LN   outputs of every LayerNorm (the A operand of q / kv / fc1 / SR-conv GEMMs; the stage outputs the head reads)
          H1   the 4x-wide hidden map between fc1 and the depthwise conv
          H2   the 4x-wide hidden map between depthwise conv + GELU and fc2
          QKV  q and kv as the attention reads them
          O    the attention output (A operand of proj)
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
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import helpers as Hh
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT
    from paif_amd.util.util import ConfusionMeter

    path = os.path.join(Hh.GOLDEN, "gq_model_b3_32x480x640.npz")
    if not os.path.exists(path):
        path = os.path.join(Hh.GOLDEN, "gp_model_b3_8x480x640.npz")
    gq = dict(np.load(path))
    n = min(args.samples, gq["pred"].shape[0])
    dev = torch.device("cuda:0")
    t = Hh.t
    m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
    S.load_formula_weights(m, head=Hh.HEAD480)
    m = m.to(dev)

    active = {"set": frozenset(), "dt": torch.float16}
    rnd = lambda x, cls: x.to(active["dt"]).to(torch.float32) if cls in active["set"] else x
    o_ln, o_dw, o_at = ops.layernorm, ops.dwconv3_bias_gelu, ops.sr_attention
    ops.layernorm = lambda *a, **k: rnd(o_ln(*a, **k), "LN")
    ops.dwconv3_bias_gelu = lambda x, *a, **k: rnd(o_dw(rnd(x, "H1"), *a, **k), "H2")

    def attn(q, kv, *a, **k):
        r = o_at(rnd(q, "QKV"), rnd(kv, "QKV"), *a, **k)
        if isinstance(r, tuple):
            return (rnd(r[0], "O"),) + tuple(r[1:])
        return rnd(r, "O")

    ops.sr_attention = attn
    # the modules captured `ops` (the module object), so the patched attributes are what they call
    table = {}

    def run(name, classes, dt=torch.float16, storage="f16"):
        active["set"], active["dt"] = frozenset(classes), dt
        ops.set_storage(storage)
        moved, preds = [], []
        for i in range(n):
            ir, vis, lab = S.make_batch(1, 480, 640, start=i)
            with torch.no_grad():
                _, seg = m(t(ir).to(dev), t(vis).to(dev))
            pred = ConfusionMeter(9, dev).update(seg, t(lab).to(dev)).cpu().numpy()[0]
            preds.append(pred)
            moved.append(int((pred != gq["pred"][i]).sum()))
        ops.set_storage("f32")
        table[name] = dict(moved=moved, agreement=1.0 - sum(moved) / (n * 307200.0), agreement_min=1.0 - max(moved) / 307200.0)
        print("%-44s agreement %.5f  min %.5f  moved %s" % (name, table[name]["agreement"], table[name]["agreement_min"], moved), flush=True)
        return preds

    run("fusion f32, seg f32", (), storage="f32")
    base = run("fusion f16, seg f32 (as shipped)", ())
    for cls in ("LN", "H1", "H2", "QKV", "O"):
        run("fusion f16, seg: only %s in fp16" % cls, (cls,))
    run("fusion f16, seg: LN + H1 + H2 in fp16", ("LN", "H1", "H2"))
    p_all = run("fusion f16, seg: LN + H1 + H2 + QKV + O in fp16", ("LN", "H1", "H2", "QKV", "O"))
    run("fusion f16, seg: H1 + H2 in fp16", ("H1", "H2"))
    run("fusion f16, seg: LN + H1 + H2 in bf16", ("LN", "H1", "H2"), dt=torch.bfloat16)
    run("fusion f32, seg: LN + H1 + H2 in fp16", ("LN", "H1", "H2"), storage="f32")
    table["_all_vs_shipped_moved"] = [int((a != b).sum()) for a, b in zip(p_all, base)]
    print("all-fp16 seg vs shipped, moved per sample:", table["_all_vs_shipped_moved"])
    table["_samples"] = n
    table["_fixture"] = os.path.basename(path)
    if args.out:
        json.dump(table, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
