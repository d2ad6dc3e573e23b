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


def test_dense_conv_dispatch_rules_on_the_host():
    """paif_conv2d_kernel_name is host-only logic (no launch): the dispatch table of DESIGN section 4, kernel by kernel."""
    import ctypes

    L = _lib.load()

    def name(kh=3, dil=1, nsrc=1, nres=0, B=8, H=480, W=640, pool=False, in_act=0, hooks=False, precision=1, cin=32, cout=32):
        d = _lib.ConvDesc()
        one = ctypes.c_void_p(16)          # any non-null pointer: nothing is dereferenced
        for i in range(3):
            d.src[i] = one if i < nsrc else None
            d.res[i] = one if i < nres else None
        d.nsrc, d.cin, d.cout, d.kh, d.dil, d.in_act, d.precision = nsrc, cin, cout, kh, dil, in_act, precision
        d.wpk, d.out = one, one
        d.pool_partial = one if pool else None
        d.aux_out = one if hooks else None
        buf = ctypes.create_string_buffer(96)
        assert L.paif_conv2d_kernel_name(ctypes.byref(d), B, H, W, buf, len(buf)) == 0
        return buf.value.decode()

    # bench shape (9600 tiles of 8 x 32)
    assert name(nsrc=1) == "conv_bf16x3_res<3, 1, 1, 4, 0>"
    assert name(nsrc=1, nres=3) == "conv_bf16x3_res<3, 1, 1, 4, 0>"
    assert name(nsrc=1, in_act=1) == "conv_bf16x3_res<3, 1, 1, 4, 0>"
    assert name(nsrc=1, pool=True) == "conv_mfma_bf16x3<3, 1, false, 0>"
    assert name(nsrc=2) == "conv_bf16x3_ms<3, 1, 2, 0>"
    assert name(nsrc=3, nres=3) == "conv_bf16x3_ms<3, 1, 3, 0>"
    assert name(nsrc=3, in_act=1) == "conv_mfma_bf16x3<3, 1, false, 0>"        # multi-source form: no input activation
    assert name(nsrc=1, hooks=True) == "conv_mfma_bf16x3<3, 1, true, 0>"       # dgrad hooks: tile-per-workgroup kernel
    assert name(kh=1, nsrc=3) == "conv_bf16x3_ws<1, 1, 0>"
    assert name(kh=1, nsrc=3, nres=1) == "conv_mfma_bf16x3<1, 1, false, 0>"
    assert name(kh=3, dil=2, nsrc=1) == "conv_mfma_bf16x3<3, 2, false, 0>"   # fp32 maps: tile-per-workgroup since round 4 (bf16 maps: below)
    assert name(kh=3, dil=2, nsrc=2) == "conv_mfma_bf16x3<3, 2, false, 0>"
    assert name(kh=7) == "conv_mfma_bf16x3<7, 1, false, 0>"
    assert name(kh=5, dil=2) == "conv_mfma_bf16x3<5, 2, false, 0>"
    # small images: fewer than 1024 / 2048 tiles -> no persistent forms
    assert name(nsrc=1, B=2, H=64, W=96) == "conv_mfma_bf16x3<3, 1, false, 0>"
    assert name(kh=1, nsrc=3, B=2, H=64, W=96) == "conv_mfma_bf16x3<1, 1, false, 0>"
    assert name(nsrc=2, B=2, H=64, W=96) == "conv_bf16x3_ms<3, 1, 2, 0>"      # the multi-source form has no size threshold
    # bf16 activation storage: the same dispatch, the storage code is the kernels' last template argument
    def name_st(st, **kw):
        d_kh = kw.get("kh", 3)
        d = _lib.ConvDesc()
        one = ctypes.c_void_p(16)
        for i in range(3):
            d.src[i] = one if i < kw.get("nsrc", 1) else None
        d.nsrc, d.cin, d.cout, d.kh, d.dil, d.precision, d.storage = kw.get("nsrc", 1), 32, 32, d_kh, kw.get("dil", 1), 1, st
        d.in_act = kw.get("in_act", 0)
        d.wpk, d.out = one, one
        buf = ctypes.create_string_buffer(96)
        assert L.paif_conv2d_kernel_name(ctypes.byref(d), 8, 480, 640, buf, len(buf)) == 0
        return buf.value.decode()

    assert name_st(1, nsrc=3) == "conv_bf16x3_ms<3, 1, 3, 1>"
    assert name_st(2, kh=1, nsrc=3) == "conv_bf16x3_ws<1, 1, 2>"
    assert name_st(1, dil=2) == "conv_bf16x3_ws<3, 2, 1>"
    assert name_st(1, dil=2, in_act=2) == "conv_bf16x3_wsr<3, 2, 1>"            # input ReLU: the composed DilConv of the bf16 forward
    assert name(kh=3, dil=2, nsrc=1, in_act=2) == "conv_mfma_bf16x3<3, 2, false, 0>"   # fp32 storage: no ReLU form of the persistent kernel
    # exact arithmetic
    assert name(precision=0) == "conv_mfma_f32<3, 1, 32, false>"
    assert name(precision=0, cin=16, cout=16) == "conv_mfma_f32<3, 1, 16, false>"


def test_streaming_guided_filter_size_rule_on_the_host():
    """paif_guided_filter_taped_fits is host-only logic (gf_stream.h make_plan): the streaming kernels address rows with wrapping 32-bit
    offsets that the buffer range check filters, so an image's 32-channel map must stay under 2^30 bytes; anything else keeps the round-1
    pair (ops.guided_filter_pair falls back by itself)."""
    L = _lib.load()
    assert L.paif_guided_filter_taped_fits(8, 480, 640) == 1
    assert L.paif_guided_filter_taped_fits(1, 10, 10) == 1
    assert L.paif_guided_filter_taped_fits(1, 9, 100) == 0 and L.paif_guided_filter_taped_fits(1, 100, 9) == 0      # H, W must exceed 2r + 1
    assert L.paif_guided_filter_taped_fits(1, 2896, 2896) == 1          # 2896^2 * 128 B = 2^30 - 2.4 MB
    assert L.paif_guided_filter_taped_fits(1, 2897, 2897) == 0
    assert L.paif_guided_filter_taped_fits(0, 480, 640) == 0
