class DID:
    """reference fusion_model/AUIF.py (DIDFuse baseline): imported at test_original.py:18, never constructed."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("fusion_model.AUIF.DID is a competitor baseline, out of scope for paif_amd (SURVEY.md section 2)")
