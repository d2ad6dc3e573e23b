class ReCoNet:
    """reference fusion_model/Reconet.py (ReCoNet baseline): imported at test_original.py:19, never constructed."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("fusion_model.Reconet.ReCoNet is a competitor baseline, out of scope for paif_amd (SURVEY.md section 2)")
