"""Input pipeline -- counterpart of the reference's TaskFusion_dataset2.py:13-107 (`prepare_data_path`, `Fusion_dataset`),
built around BYTES: files are decoded once to uint8 (PIL only, no cv2), batches travel to the GPU as pinned uint8 and the
reference loader's float tensors are formed on the device (csrc/io_kernels.hip) -- 4x less PCIe traffic than uploading the
float32 tensors the reference builds on the host.

Contract of `Fusion_dataset.__getitem__` (what the reference's DataLoader yields, :74-104), kept for drop-in use:
  (vis float32 [3,H,W] = RGB/255, ir float32 [1,H,W] = gray/255, label int64 [H,W], file name of the visible image)
`cv2.imread(path, 0)` is the grey-scale read: single-channel files are returned as stored; colour files go through
OpenCV's fixed-point BT.601 conversion, restated here so the bytes match:  gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14.
(16-bit or palette infrared files are outside what the reference's scripts feed it and are rejected.)"""
import glob
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data.dataset import Dataset

_EXTENSIONS = ("bmp", "tif", "jpg", "png")          # the four globs of TaskFusion_dataset2.py:16-19, in that order
_TRAIN_DIRS = dict(vi_path='./MSRS/Visible/train/MSRS/', ir_path='./MSRS/Infrared/train/MSRS/', label_path='./MSRS/Label/train/MSRS/')


def prepare_data_path(dataset_path):
    """-> (sorted image paths with one of the four extensions, sorted directory listing)   (TaskFusion_dataset2.py:13-22)"""
    paths = sorted(p for ext in _EXTENSIONS for p in glob.glob(os.path.join(dataset_path, "*." + ext)))
    return paths, sorted(os.listdir(dataset_path))


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
    """Directory triple (visible, infrared, label) -> items.  split 'train' uses the reference's hard-coded MSRS directories
    (:31-33), 'val' the three given ones; 'test' constructs an empty object exactly like the reference (whose __len__ then
    raises AttributeError)."""

    def __init__(self, split, ir_path=None, vi_path=None, label_path=None):
        super().__init__()
        assert split in ['train', 'val', 'test'], 'split must be "train"|"val"|"test"'
        if split == 'test':
            return
        dirs = _TRAIN_DIRS if split == 'train' else dict(vi_path=vi_path, ir_path=ir_path, label_path=label_path)
        self.split = split
        self.filepath_vis, self.filenames_vis = prepare_data_path(dirs["vi_path"])
        self.filepath_ir, self.filenames_ir = prepare_data_path(dirs["ir_path"])
        self.filepath_label, self.filenames_label = prepare_data_path(dirs["label_path"])
        self.length = min(len(self.filenames_vis), len(self.filenames_ir))

    def decode(self, index):
        """The item as decoded bytes: (vis uint8 [H,W,3], ir uint8 [H,W], label array [H,W] as stored, name)."""
        vis = np.array(Image.open(self.filepath_vis[index]))
        if vis.dtype != np.uint8 or vis.ndim != 3:
            raise ValueError("visible image %s: expected 8-bit RGB" % self.filepath_vis[index])
        return vis[..., :3], imread_gray(self.filepath_ir[index]), np.array(Image.open(self.filepath_label[index])), self.filenames_vis[index]

    def __getitem__(self, index):
        vis, ir, label, name = self.decode(index)
        scale = np.float32(255.0)
        return (torch.from_numpy(np.ascontiguousarray(vis.transpose(2, 0, 1)).astype(np.float32) / scale),
                torch.from_numpy(ir.astype(np.float32)[None] / scale),
                torch.from_numpy(label.astype(np.int64)), name)

    def __len__(self):
        return self.length


class _ByteItems(Dataset):
    """The same items as uint8 tensors (what `device_batches` moves over PCIe)."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        vis, ir, label, name = self.ds.decode(index)
        if label.dtype != np.uint8:
            label = label.astype(np.int64)          # rare label encodings: widened on the host
        return torch.from_numpy(np.ascontiguousarray(vis)), torch.from_numpy(ir), torch.from_numpy(label), name


def device_batches(dataset, device, batch_size=1, num_workers=0, with_names=False):
    """Yields (vis [B,3,H,W], ir [B,1,H,W], label int64 [B,H,W]) device tensors -- the `batches` argument of
    paif_amd.harness.* (batch_size 1 = the reference's test_original.py:111).  Decode runs in DataLoader workers; the
    uint8 host batch is pinned and uploaded on a side stream one batch ahead; the consumer waits on THAT batch's event
    only (so compute on batch i overlaps the upload of batch i+1) and the /255, layout and int64 conversions run in HIP
    kernels on the compute stream."""
    from torch.utils.data import DataLoader
    from . import ops
    loader = DataLoader(_ByteItems(dataset), batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True,
                        drop_last=False)
    side = torch.cuda.Stream(device=device)

    def upload(item):
        vis, ir, label, names = item
        with torch.cuda.stream(side):
            dev = tuple(x.to(device, non_blocking=True) for x in (vis, ir, label))
            done = torch.cuda.Event()
            done.record(side)
        return dev, done, list(names)

    def ready(job):
        (vis, ir, label), done, names = job
        cur = torch.cuda.current_stream(device)
        cur.wait_event(done)
        for x in (vis, ir, label):
            x.record_stream(cur)
        out = (ops.u8_to_planes(vis), ops.u8_to_planes(ir), ops.u8_to_i64(label) if label.dtype == torch.uint8 else label)
        return out + (names,) if with_names else out

    pending = None
    for item in loader:
        job = upload(item)
        if pending is not None:
            yield ready(pending)
        pending = job
    if pending is not None:
        yield ready(pending)
