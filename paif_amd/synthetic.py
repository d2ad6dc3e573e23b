"""Deterministic synthetic inputs and formula weights (numpy only, portable).

No checkpoint or dataset of the reference is available (reference README.md:34-43 gives
Drive links only), so every parity case runs on inputs and weights produced by the
integer formulas below.  The same functions are used by the oracle's golden-vector
script (this container), by the GPU parity tests and by bench.py (GPU box), so both
sides see bit-identical float32 data without shipping any tensor files.

Input contract mirrored from the reference dataset class (TaskFusion_dataset2.py:74-104):
vis = RGB uint8 / 255 -> float32 [3,H,W]; ir = gray uint8 / 255 -> float32 [1,H,W];
label = int64 [H,W] in {0..8} (255 = ignore).
"""
import os
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed, n, offset=0):
    """n float64 values in [0,1) from a counter-based hash of (seed, index)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + idx)
    return (key >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def key_seed(name):
    """Stable 32-bit seed of a state_dict key (crc32 of its utf-8 name)."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def formula_tensor(name, shape, salt=0):
    """Formula value for one state_dict entry, chosen by the key's suffix.

    * ``num_batches_tracked``           -> 0 (int64)
    * ``running_var``                   -> uniform [0.5, 1.5)
    * ``running_mean``                  -> uniform [-0.1, 0.1)
    * norm ``weight`` (1-D, LayerNorm/BatchNorm)  -> uniform [0.8, 1.2)
    * PReLU ``weight`` (shape (1,))     -> uniform [0.1, 0.3)
    * ``bias``                          -> uniform [-0.05, 0.05)
    * conv / linear ``weight``          -> uniform [-1, 1) * sqrt(3 / fan_in)   (unit-gain)
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    seed = (key_seed(name) + 0x9E3779B1 * salt) & 0xFFFFFFFF
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    u = hash_uniform(seed, n)
    if name.endswith("running_var"):
        v = 0.5 + u
    elif name.endswith("running_mean"):
        v = (u - 0.5) * 0.2
    elif name.endswith("bias"):
        v = (u - 0.5) * 0.1
    elif len(shape) == 1 and shape[0] == 1:
        v = 0.1 + 0.2 * u  # PReLU slope
    elif len(shape) == 1:
        v = 0.8 + 0.4 * u  # norm gain
    else:
        fan_in = int(np.prod(shape[1:]))
        v = (2.0 * u - 1.0) * np.sqrt(3.0 / fan_in)
    return v.astype(np.float32).reshape(shape)


def formula_state_dict(shapes, salt=0):
    """shapes: mapping key -> shape.  Returns key -> numpy array."""
    return {k: formula_tensor(k, s, salt) for k, s in shapes.items()}


# Calibrated segmentation head ---------------------------------------------------------------------------------------
# With formula weights everywhere the SegFormer head predicts ONE class on every pixel (the class means of its last layer
# swamp the spatial variation), which makes argmax / confusion-matrix / mIoU parity checks vacuous.  oracle/calibrate_head.py
# therefore fits `decoder.linear_pred.{weight,bias}` (core/segformer_head.py:57) per parity case by ridge regression of the
# synthetic labels on the REFERENCE's own head feature and stores the float32 table in synthetic_head.npz: with it the
# reference's argmax has all 9 classes (>= 5 % each), near-ties along every class boundary and a label-correlated map.
HEAD_KEYS = ("decoder.linear_pred.weight", "decoder.linear_pred.bias")
_HEAD_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic_head.npz")
_head_cache = {}


def head_tag(backbone, B, H, W):
    """Name of a calibration case: the backbone and the batch it was fitted on, e.g. 'mit_b3_1x480x640'."""
    return "%s_%dx%dx%d" % (backbone, B, H, W)


def calibrated_head(tag):
    """(weight [9,256,1,1], bias [9]) float32 of a calibration case, plus the reference's class shares on its inputs."""
    if not _head_cache:
        _head_cache.update(dict(np.load(_HEAD_FILE)))
    if tag + ".weight" not in _head_cache:
        raise KeyError("no calibrated head %r (have: %s)" % (tag, sorted(k[:-7] for k in _head_cache if k.endswith(".weight"))))
    return _head_cache[tag + ".weight"], _head_cache[tag + ".bias"], _head_cache[tag + ".class_share"]


def apply_head(state_dict, tag):
    """Overwrite the `...decoder.linear_pred.{weight,bias}` entries of a (numpy or torch) state_dict with a calibrated head."""
    w, b, _ = calibrated_head(tag)
    hit = 0
    for k in list(state_dict):
        for suffix, val in zip(HEAD_KEYS, (w, b)):
            if k.endswith(suffix):
                old = state_dict[k]
                if isinstance(old, np.ndarray):
                    state_dict[k] = val.copy()
                else:
                    import torch

                    state_dict[k] = torch.from_numpy(val.copy()).to(old.dtype)
                hit += 1
    if hit != 2:
        raise KeyError("state_dict has no decoder.linear_pred entries to calibrate")
    return state_dict


def load_formula_weights(module, salt=0, strict=True, head=None):
    """Fill a torch module's state_dict from the formula (strict key check); `head` = a calibration tag (head_tag(...))
    replaces the segmentation head's last layer by the fitted table."""
    import torch

    sd = module.state_dict()
    new = {k: torch.from_numpy(formula_tensor(k, tuple(v.shape), salt)).to(v.dtype) for k, v in sd.items()}
    if head is not None:
        apply_head(new, head)
    module.load_state_dict(new, strict=strict)
    return module


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------
def _smooth_field(seed, H, W, freqs=((1, 1), (2, 3), (5, 2), (3, 7))):
    """Natural-image-like smooth uint8 field + hash noise (guided-filter sensitivity grows in
    flat regions, so plain white noise would be the wrong test signal)."""
    yy = np.arange(H, dtype=np.float64)[:, None] / H
    xx = np.arange(W, dtype=np.float64)[None, :] / W
    ph = hash_uniform(seed, 3 * len(freqs))
    f = np.zeros((H, W), dtype=np.float64)
    for i, (fy, fx) in enumerate(freqs):
        amp = 0.5 + ph[3 * i]
        f += amp * np.sin(2 * np.pi * (fy * yy + ph[3 * i + 1])) * np.cos(2 * np.pi * (fx * xx + ph[3 * i + 2]))
    f = (f - f.min()) / (f.max() - f.min() + 1e-12)
    # blocky objects (edges) + noise
    by = (np.arange(H)[:, None] * 7 // max(H, 1)) % 2
    bx = (np.arange(W)[None, :] * 9 // max(W, 1)) % 2
    f = 0.70 * f + 0.18 * (by ^ bx)
    noise = hash_uniform(seed ^ 0x5BD1E995, H * W).reshape(H, W)
    f = f + 0.12 * (noise - 0.5)
    f = np.clip(f, 0.0, 1.0)
    return np.floor(f * 255.0 + 0.5).astype(np.uint8)


def make_pair(index, H=480, W=640):
    """One (ir [1,H,W], vis [3,H,W]) float32 pair in [0,1] = uint8/255, seeded by sample index."""
    ir = _smooth_field(1000 + 17 * index, H, W).astype(np.float32) / np.float32(255.0)
    vis = np.stack(
        [_smooth_field(2000 + 17 * index + c, H, W, freqs=((1, 2), (3, 1), (4, 5), (2, 6))) for c in range(3)], 0
    ).astype(np.float32) / np.float32(255.0)
    return ir[None], vis


def make_label(index, H=480, W=640, n_class=9, ignore=255):
    """Blocky 9-class label map, a few px = 255 to exercise ignore_index."""
    yy = np.arange(H)[:, None]
    xx = np.arange(W)[None, :]
    by = yy * 6 // H
    bx = xx * 8 // W
    lab = ((by * 5 + bx * 3 + index) % n_class).astype(np.int64)
    # a diagonal stripe of ignore pixels
    lab[((yy + xx + index) % 97) == 0] = ignore
    return lab


def make_batch(B, H=480, W=640, start=0):
    """Returns ir [B,1,H,W], vis [B,3,H,W] float32, label [B,H,W] int64."""
    irs, viss, labs = [], [], []
    for i in range(start, start + B):
        ir, vis = make_pair(i, H, W)
        irs.append(ir)
        viss.append(vis)
        labs.append(make_label(i, H, W))
    return np.stack(irs), np.stack(viss), np.stack(labs)


def make_delta0(index, shape, epsilon):
    """PGD start perturbation: counter-hash uniform in [-eps, eps) (replaces the reference's
    global-RNG ``uniform_`` at attack/attack.py:434,439 so both sides start identically)."""
    n = int(np.prod(shape))
    u = hash_uniform(0xD17A0000 + index, n)
    return ((2.0 * u - 1.0) * epsilon).astype(np.float32).reshape(shape)


def make_feature(seed, shape, lo=-1.0, hi=1.0):
    """Generic float32 feature tensor for per-operator tests."""
    n = int(np.prod(shape))
    return (lo + (hi - lo) * hash_uniform(seed, n)).astype(np.float32).reshape(shape)


def make_smooth_feature(seed, B, C, H, W):
    """Smooth multi-channel feature (uint8-quantised field per channel, affine-mixed)."""
    out = np.zeros((B, C, H, W), dtype=np.float32)
    for b in range(B):
        for c in range(C):
            f = _smooth_field(seed + 131 * b + 7 * c, H, W).astype(np.float32) / np.float32(255.0)
            out[b, c] = f * np.float32(0.5 + 0.03 * c) - np.float32(0.1 * (c % 5))
    return out


def sample_indices(n, k=512):
    """Up to k evenly spaced flat indices into a tensor of n elements (golden fixtures keep a sample of the big
    gradient / parameter tensors, not all 4 M values)."""
    if n <= k:
        return np.arange(n, dtype=np.int64)
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))
