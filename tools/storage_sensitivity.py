"""Per-map sensitivity of the segmentation argmax to 16-bit STORAGE of the fusion network's maps.

Emulation on the fp32-storage kernels: after the producing launch a map is rounded IN PLACE to the 16-bit format (bf16 or fp16,
round-to-nearest-even) -- exactly what a 16-bit store in that kernel's epilogue would keep (the accumulation order is the fp32
kernel's).  One class of maps at a time ("only X") and everything but one class ("all but X"), weights exact or rounded to the
format, against the reference's 1x480x640 mit_b3 golden (calibrated head).  Prints / writes a JSON table:
logit error (mean / max, of the logit range), moved pixels of the x4-upsampled argmax, mIoU delta.

    python tools/storage_sensitivity.py [--out gpurun_out/storage_sensitivity.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from paif_amd import ops  # noqa: E402
from paif_amd import synthetic as S  # noqa: E402

# classes of 32-channel maps of the inference forward (core/model_fusion_auto.py:625-635 and the cells behind it)
CLASSES = ["stem_twin", "gf_lf", "decomp1x1", "rdb_x1x2", "rdb_out", "stream_out", "blend", "eca_r", "eca_o", "eca_out", "res7x7", "feature2"]
# conv2d launches of one fp32-storage inference forward, in launch order (checked against the logged descriptors below)
CONV_CLASS = ["decomp1x1", "decomp1x1",
              "rdb_x1x2", "rdb_x1x2", "rdb_out", "stream_out",            # infrared chain: RDB, DilConv (+ lf + fir)
              "rdb_x1x2", "rdb_x1x2", "rdb_out", "rdb_x1x2", "rdb_x1x2", "stream_out",   # visible chain2: RDB, RDB (+ hf + fvis)
              "eca_r", "eca_o", "res7x7", "feature2"]
CONV_SHAPE = [(1, 1, 3, 0), (1, 1, 3, 0),
              (3, 1, 1, 0), (3, 1, 2, 0), (3, 1, 3, 1), (3, 2, 1, 3),
              (3, 1, 1, 0), (3, 1, 2, 0), (3, 1, 3, 1), (3, 1, 1, 0), (3, 1, 2, 0), (3, 1, 3, 3),
              (3, 1, 1, 0), (3, 1, 1, 0), (7, 1, 1, 0), (3, 2, 1, 2)]


def rnd_(x, fmt):
    if fmt == "bf16":
        x.copy_(x.to(torch.bfloat16))
    elif fmt == "f16":
        x.copy_(x.to(torch.float16))
    return x


class Emu:
    """Monkeypatches the ops entry points of the inference forward."""

    def __init__(self):
        self.on = {}        # class -> format
        self.wfmt = None    # weight format or None
        self.n = 0
        self.twins = {}
        self.log = []
        self._orig = dict(conv2d=ops.conv2d, stem=ops.stem, gf=ops.guided_filter_pair, blend=ops.spa_blend, eca=ops.eca_finish,
                          pack=ops.pack_conv_weight, pack1=ops.pack_decomp1x1_weight)
        ops.conv2d, ops.stem, ops.guided_filter_pair = self.conv2d, self.stem, self.gf
        ops.spa_blend, ops.eca_finish = self.blend, self.eca
        ops.pack_conv_weight, ops.pack_decomp1x1_weight = self.pack, self.pack1

    def begin(self):
        self.n = 0
        self.twins.clear()
        self.log = []

    def _tw(self, x):
        return self.twins.get(x.data_ptr(), x)

    def stem(self, *a, **k):
        feat, guide = self._orig["stem"](*a, **k)
        f = self.on.get("stem_twin")
        if f:   # the guided filter keeps the fp32 map; the 1x1 and the chain's outer residual take the 16-bit twin
            self.twins[feat.data_ptr()] = rnd_(feat.clone(), f)
        return feat, guide

    def gf(self, *a, **k):
        lf = self._orig["gf"](*a, **k)
        f = self.on.get("gf_lf")
        return rnd_(lf, f) if f else lf

    def conv2d(self, srcs, wpk, kh, dil=1, **k):
        res = tuple(r for r in k.get("res", ()) if r is not None)
        cls = CONV_CLASS[self.n]
        assert (kh, dil, len(srcs), len(res)) == CONV_SHAPE[self.n], (self.n, kh, dil, len(srcs), len(res))
        self.n += 1
        self.log.append((kh, dil, len(srcs), len(res)))
        srcs = [self._tw(s) for s in srcs]
        k["res"] = tuple(self._tw(r) for r in res)
        r = self._orig["conv2d"](srcs, wpk, kh, dil, **k)
        f = self.on.get(cls)
        if f:
            rnd_(r[0] if isinstance(r, tuple) else r, f)
        return r

    def blend(self, *a, **k):
        r = self._orig["blend"](*a, **k)
        f = self.on.get("blend")
        return rnd_(r, f) if f else r

    def eca(self, *a, **k):
        r = self._orig["eca"](*a, **k)
        f = self.on.get("eca_out")
        return rnd_(r, f) if f else r

    def _w(self, w):
        return rnd_(w.detach().clone(), self.wfmt) if self.wfmt else w

    def pack(self, w, *a, **k):
        return self._orig["pack"](self._w(w), *a, **k)

    def pack1(self, w, *a, **k):
        return self._orig["pack1"](self._w(w), *a, **k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import helpers as Hh
    from paif_amd.core.model_fusion_auto import Network_MM_Searched
    from paif_amd.genotypes import FUSION_AT
    from paif_amd.util.util import ConfusionMeter, compute_results

    g = dict(np.load(os.path.join(Hh.GOLDEN, "gf_model_b3_1x480x640.npz")))
    dev = torch.device("cuda:0")
    t = Hh.t
    ops.CONFIG["stem_out_fused_f32"] = True
    ir, vis, lab = S.make_batch(1, 480, 640)
    irt, vist, labt = t(ir).to(dev), t(vis).to(dev), t(lab).to(dev)
    rng = float(g["logits"].max() - g["logits"].min())
    miou_ref = float(np.nanmean(compute_results(g["conf"])[2]))
    emu = Emu()

    def fresh_model():
        m = Network_MM_Searched(32, FUSION_AT, None, None, "mit_b3", num_classes=9).eval()
        S.load_formula_weights(m, head=Hh.HEAD480)
        return m.to(dev)

    def run(on, wfmt=None):
        emu.on, emu.wfmt = dict(on), wfmt
        m = fresh_model()          # weight packs are cached per model: a fresh one per weight format
        emu.begin()
        with torch.no_grad():
            fused, seg = m(irt, vist)
        meter = ConfusionMeter(9, dev)
        pred = meter.update(seg, labt).cpu().numpy()
        lerr = (seg.cpu() - t(g["logits"])).abs()
        d64 = (fused.cpu().double() - t(g["fused64"]).double()).abs()
        dis = pred != g["pred"]
        miou = float(np.nanmean(compute_results(meter.conf.cpu().numpy())[2]))
        return dict(moved=int(dis.sum()), agree=float(1.0 - dis.mean()), logits_mean=float(lerr.mean()) / rng, logits_max=float(lerr.max()) / rng,
                    fused_mean=float(d64.mean()), fused_max=float(d64.max()), miou_delta_pt=100.0 * (miou - miou_ref))

    table = {}

    def rec(name, on, wfmt=None):
        table[name] = r = run(on, wfmt)
        print("%-44s moved %6d  agree %.5f  logits mean %.2e max %.2e  fused mean %.2e max %.2e  dmIoU %+.3f pt" % (
            name, r["moved"], r["agree"], r["logits_mean"], r["logits_max"], r["fused_mean"], r["fused_max"], r["miou_delta_pt"]), flush=True)

    rec("f32 (no rounding)", {})
    assert emu.log == CONV_SHAPE, emu.log
    for fmt in ("bf16", "f16"):
        rec("weights only %s" % fmt, {}, fmt)
        rec("all maps %s, weights exact" % fmt, {c: fmt for c in CLASSES})
        rec("all maps %s, weights %s" % (fmt, fmt), {c: fmt for c in CLASSES}, fmt)
        for c in CLASSES:
            rec("only %s in %s" % (c, fmt), {c: fmt})
        for c in CLASSES:
            rec("all but %s in %s (weights %s)" % (c, fmt, fmt), {k: fmt for k in CLASSES if k != c}, fmt)
    # mixed plans: fp16 everywhere except the most sensitive classes (filled in from the table above)
    order = sorted(CLASSES, key=lambda c: -table["only %s in f16" % c]["moved"])
    print("f16 sensitivity order:", order)
    for keep in (1, 2, 3, 4):
        kept = order[:keep]
        rec("f16 maps+weights, fp32 kept: %s" % "+".join(kept), {k: "f16" for k in CLASSES if k not in kept}, "f16")
    table["_meta"] = dict(logit_range=rng, miou_reference=miou_ref, golden="gf_model_b3_1x480x640", f16_order=order,
                          reference_f32_vs_f64_pixels=int((g["pred"] != g["pred64"]).sum()))
    if args.out:
        json.dump(table, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
