"""CPU: the C-ABI library loads and exports every symbol include/paif_hip.h declares (no compute calls)."""
import os
import re

from paif_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "paif_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(paif_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from paif_amd import build

    build.build()
    L = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libpaif_hip.so lacks %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    assert L.paif_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    import pytest

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpaif_hip.so")
    with pytest.raises(_lib.PaifLibraryError):
        _lib.load()


def test_cpu_tensor_is_refused():
    import pytest
    import torch

    from paif_amd import ops

    with pytest.raises(RuntimeError):
        ops.rgb2ycrcb(torch.zeros(1, 3, 16, 16))
