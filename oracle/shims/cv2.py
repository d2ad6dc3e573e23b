"""Oracle shim for cv2 (absent from the image).  attack/attack.py:8 only imports it; TaskFusion_dataset2.py:57 calls
`cv2.imread(path, 0)` -- the grey-scale read -- which is restated here from OpenCV's documented behaviour: single-channel
8-bit files are returned as stored, colour files are converted with the fixed-point BT.601 weights
gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14.  Test infrastructure only."""
import numpy as np
from PIL import Image


def imread(path, flags=1):
    if flags != 0:
        raise NotImplementedError("cv2 shim: only imread(path, 0) is restated")
    im = Image.open(path)
    if im.mode == "L":
        return np.array(im)
    a = np.array(im.convert("RGB")).astype(np.uint32)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
