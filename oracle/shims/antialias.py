"""Oracle shim: placeholder for the missing `antialias` module (operations_m.py:4; only the
unused ResidualDownSample touches it)."""
import torch.nn as nn


class Downsample(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
