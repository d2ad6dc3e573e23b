"""Input pipeline -- counterpart of the reference's TaskFusion_dataset2.py:13-107 (`prepare_data_path`, `Fusion_dataset`)
without the cv2 dependency, plus a pinned-memory prefetcher that hands the evaluation harnesses device tensors.

Contract reproduced (val split, and train with the reference's hard-coded MSRS directories):
  item = (vis float32 [3,H,W] = RGB/255, ir float32 [1,H,W] = gray/255, label int64 [H,W], file name of the visible image)
`cv2.imread(path, 0)` is the grey-scale read: single-channel files are returned as stored; colour files go through
OpenCV's fixed-point BT.601 conversion, restated here so the bytes match:  gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14.
(16-bit or palette infrared files are outside what the reference's scripts feed it and are rejected.)"""
import glob
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data.dataset import Dataset


def prepare_data_path(dataset_path):
    """TaskFusion_dataset2.py:13-22: sorted paths of *.bmp/*.tif/*.jpg/*.png and the sorted directory listing."""
    filenames = os.listdir(dataset_path)
    data = glob.glob(os.path.join(dataset_path, "*.bmp"))
    data.extend(glob.glob(os.path.join(dataset_path, "*.tif")))
    data.extend(glob.glob(os.path.join(dataset_path, "*.jpg")))
    data.extend(glob.glob(os.path.join(dataset_path, "*.png")))
    data.sort()
    filenames.sort()
    return data, filenames


def imread_gray(path):
    """cv2.imread(path, 0) for 8-bit files."""
    im = Image.open(path)
    if im.mode == "L":
        return np.array(im)
    if im.mode in ("RGB", "RGBA"):
        a = np.array(im.convert("RGB")).astype(np.uint32)
        return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    raise ValueError("imread_gray: unsupported image mode %r in %s (expected 8-bit grey or RGB)" % (im.mode, path))


class Fusion_dataset(Dataset):
    def __init__(self, split, ir_path=None, vi_path=None, label_path=None):
        super().__init__()
        assert split in ['train', 'val', 'test'], 'split must be "train"|"val"|"test"'
        if split == 'train':
            vi_path, ir_path, label_path = './MSRS/Visible/train/MSRS/', './MSRS/Infrared/train/MSRS/', './MSRS/Label/train/MSRS/'
        if split in ('train', 'val'):
            self.filepath_vis, self.filenames_vis = prepare_data_path(vi_path)
            self.filepath_ir, self.filenames_ir = prepare_data_path(ir_path)
            self.filepath_label, self.filenames_label = prepare_data_path(label_path)
            self.split = split
            self.length = min(len(self.filenames_vis), len(self.filenames_ir))
        # split == 'test': the reference's constructor sets nothing either (its __len__ then raises AttributeError)

    def __getitem__(self, index):
        image_vis = np.array(Image.open(self.filepath_vis[index]))
        image_inf = imread_gray(self.filepath_ir[index])
        label = np.array(Image.open(self.filepath_label[index]))
        image_vis = np.asarray(Image.fromarray(image_vis), dtype=np.float32).transpose((2, 0, 1)) / 255.0
        image_ir = np.expand_dims(np.asarray(Image.fromarray(image_inf), dtype=np.float32) / 255.0, axis=0)
        label = np.asarray(Image.fromarray(label), dtype=np.int64)
        return torch.tensor(image_vis), torch.tensor(image_ir), torch.tensor(label), self.filenames_vis[index]

    def __len__(self):
        return self.length


def device_batches(dataset, device, batch_size=1, num_workers=0, with_names=False):
    """Yields (vis [B,3,H,W], ir [B,1,H,W], label int64 [B,H,W]) device tensors -- the `batches` argument of
    paif_amd.harness.* (batch_size 1 = the reference's test_original.py:111).  Decode runs in DataLoader workers, the
    host batch is pinned and the upload of batch i+1 is issued on a side stream while batch i is being consumed."""
    from torch.utils.data import DataLoader
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True, drop_last=False)
    side = torch.cuda.Stream(device=device)
    nxt = None

    def upload(item):
        vis, ir, label, names = item
        with torch.cuda.stream(side):
            out = (vis.to(device, non_blocking=True), ir.to(device, non_blocking=True), label.to(device, non_blocking=True))
        return out, names

    for item in loader:
        cur, nxt = nxt, upload(item)
        if cur is not None:
            yield _ready(cur, side, with_names)
    if nxt is not None:
        yield _ready(nxt, side, with_names)


def _ready(pair, side, with_names):
    (vis, ir, label), names = pair
    torch.cuda.current_stream().wait_stream(side)
    for x in (vis, ir, label):
        x.record_stream(torch.cuda.current_stream())
    return (vis, ir, label, list(names)) if with_names else (vis, ir, label)
