"""Input pipeline: paif_amd.TaskFusion_dataset2 against the reference's class on a
directory of synthetic PNGs (the reference is imported when /root/reference is present; cv2 comes from oracle/shims)."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from paif_amd import synthetic as S
from paif_amd.TaskFusion_dataset2 import Fusion_dataset, prepare_data_path, imread_gray

REF = "/root/reference"


def _write_set(root, n=3, H=48, W=64, colour_ir=False):
    for d in ("vi", "ir", "label"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for i in range(n):
        ir, vis = S.make_pair(i, H, W)
        lab = S.make_label(i, H, W)
        Image.fromarray(np.uint8(np.round(vis.transpose(1, 2, 0) * 255))).save(os.path.join(root, "vi", "%05dD.png" % i))
        g = np.uint8(np.round(ir[0] * 255))
        im = Image.fromarray(np.stack([g, g // 2, 255 - g], -1)) if colour_ir else Image.fromarray(g)
        im.save(os.path.join(root, "ir", "%05dD.png" % i))
        Image.fromarray(np.uint8(lab)).save(os.path.join(root, "label", "%05dD.png" % i))
    return os.path.join(root, "ir"), os.path.join(root, "vi"), os.path.join(root, "label")


def test_item_contract(tmp_path):
    irp, vip, lbp = _write_set(str(tmp_path))
    ds = Fusion_dataset('val', ir_path=irp, vi_path=vip, label_path=lbp)
    assert len(ds) == 3
    vis, ir, lab, name = ds[1]
    assert vis.dtype == torch.float32 and tuple(vis.shape) == (3, 48, 64) and 0.0 <= float(vis.min()) and float(vis.max()) <= 1.0
    assert ir.dtype == torch.float32 and tuple(ir.shape) == (1, 48, 64)
    assert lab.dtype == torch.int64 and tuple(lab.shape) == (48, 64)
    assert name == "00001D.png"
    ir0, vis0 = S.make_pair(1, 48, 64)
    assert np.array_equal(np.uint8(np.round(ir0[0] * 255)), np.uint8(np.round(ir.numpy()[0] * 255)))
    paths, names = prepare_data_path(vip)
    assert names == sorted(names) and [os.path.basename(p) for p in paths] == names
    with pytest.raises(AssertionError):
        Fusion_dataset('bogus')


def test_grey_read_of_colour_file_uses_opencv_weights(tmp_path):
    p = str(tmp_path / "c.png")
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], dtype=np.uint8)
    Image.fromarray(rgb).save(p)
    assert imread_gray(p).tolist() == [[76, 150, 29, 124]]     # cv2.cvtColor(..., COLOR_BGR2GRAY) on these pixels


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("colour_ir", [False, True])
def test_matches_reference_class(tmp_path, colour_ir):
    irp, vip, lbp = _write_set(str(tmp_path), colour_ir=colour_ir)
    shims = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shims")
    sys.path[:0] = [shims, REF]
    try:
        sys.modules.pop("TaskFusion_dataset2", None)
        import TaskFusion_dataset2 as R
        ref = R.Fusion_dataset('val', ir_path=irp, vi_path=vip, label_path=lbp)
    finally:
        sys.path.remove(shims)
        sys.path.remove(REF)
        sys.modules.pop("TaskFusion_dataset2", None)
        sys.modules.pop("cv2", None)
    mine = Fusion_dataset('val', ir_path=irp, vi_path=vip, label_path=lbp)
    assert len(mine) == len(ref)
    for i in range(len(ref)):
        a, b = mine[i], ref[i]
        for x, y in zip(a[:3], b[:3]):
            assert x.dtype == y.dtype and torch.equal(x, y)
        assert a[3] == b[3]
