"""TEST INFRASTRUCTURE (oracle side) -- imports the real reference from /root/reference.

Only usable in the build container (the reference does not exist on the GPU box).  Used by
oracle/make_golden.py to emit golden vectors and by tests/test_oracle_vs_reference.py to pin
the oracle's restatement against the reference itself, whenever /root/reference is present.

The reference does not import as shipped (SURVEY.md headline fact 2): seven third-party
modules are missing and RGB2YCrCb/YCrCb2RGB/attack_* call .cuda() unconditionally
(core/model_fusion_auto.py:81,98-100; attack/attack.py:423,433,438).  This module puts
oracle/shims first on sys.path and patches .cuda() to identity.  Nothing here is product code.
"""
import contextlib
import io
import os
import sys

REF_ROOT = "/root/reference"
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")
_loaded = {}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "core"))


def load():
    """Returns a dict of reference modules: mfa (core.model_fusion_auto), ops (operations_m),
    attack (attack.attack), mit (core.mix_transformer), head, loss, ssim, util, optimizer."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch

    sys.dont_write_bytecode = True  # reference dir is read-only
    # our repo ships drop-in packages named like the reference's (core, attack, util, utils,
    # operations_m); make sure the REAL reference wins inside this process.
    for name in list(sys.modules):
        root = name.split(".")[0]
        if root in ("core", "attack", "util", "utils", "operations_m", "pytorch_ssim"):
            del sys.modules[name]
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, _SHIMS)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    with contextlib.redirect_stdout(io.StringIO()):
        import operations_m as ops
        import core.model_fusion_auto as mfa
        import core.mix_transformer as mit
        import core.segformer_head as head
        import core.loss as loss
        import attack.attack as attack
        import pytorch_ssim as ssim
        import util.util as util
        import utils.optimizer as optimizer
    assert ops.__file__.startswith(REF_ROOT), ops.__file__
    assert mfa.__file__.startswith(REF_ROOT), mfa.__file__
    _loaded.update(dict(ops=ops, mfa=mfa, mit=mit, head=head, loss=loss, attack=attack, ssim=ssim,
                        util=util, optimizer=optimizer))
    return _loaded


@contextlib.contextmanager
def quiet():
    """Reference constructors print every primitive (core/model_fusion_auto.py:411,433)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield
