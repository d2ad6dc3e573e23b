from paif_amd.core.mix_transformer import *  # noqa: F401,F403
from paif_amd.core.mix_transformer import mit_b0, mit_b1, mit_b2, mit_b3, mit_b4, mit_b5  # noqa: F401
