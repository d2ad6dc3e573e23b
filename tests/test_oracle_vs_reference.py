"""CPU, build container only: the oracle against the LIVE reference import on ragged sizes that are
not in the golden set.  Skipped on the GPU box (no /root/reference there)."""
import pytest
import torch

from oracle import paif_oracle as O
from oracle import ref_import
from paif_amd import synthetic as S
from tests.helpers import t, maxabs

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")


@pytest.mark.parametrize("B,H,W", [(1, 40, 56), (3, 32, 72)])
def test_model_forward_live(B, H, W):
    R = ref_import.load()
    with ref_import.quiet():
        m = R["mfa"].Network_MM_Searched(32, O.FUSION_AT, None, None, "mit_b0", num_classes=9)
    m.eval()
    S.load_formula_weights(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ir, vis, _ = S.make_batch(B, H, W, start=5)
    with torch.no_grad():
        f_ref, s_ref = m(t(ir), t(vis))
        f, s = O.model_forward(t(ir), t(vis), sd, "mit_b0")
    assert maxabs(f, f_ref) <= 1e-5
    assert maxabs(s, s_ref) <= 1e-4


def test_guided_filter_too_small_asserts_like_reference():
    ref_import.load()
    from guided_filter_pytorch.guided_filter import GuidedFilter

    with pytest.raises(AssertionError):
        GuidedFilter(4, 1e-3)(torch.zeros(1, 1, 9, 20), torch.zeros(1, 4, 9, 20))
