from paif_amd.core.segformer_head import MLP, SegFormerHead  # noqa: F401
