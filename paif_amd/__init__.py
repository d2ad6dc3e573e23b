"""paif_amd -- the PAIF hot path (fusion network -> SegFormer -> PGD loop -> adversarial-training step) on MI355X / gfx950:
the reference's nn.Module / function API over hand-written HIP kernels (include/paif_hip.h, paif_amd/lib/libpaif_hip.so)."""


def invalidate_weight_caches():
    """See paif_amd.operations_m.invalidate_weight_caches."""
    from .operations_m import invalidate_weight_caches as _f
    _f()
