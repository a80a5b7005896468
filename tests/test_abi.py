"""CPU-only: the C-ABI library builds, loads and exports every symbol include/addvisor_hip.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from addvisor_hip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "addvisor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(advh_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    return _lib.build()


def test_header_matches_bindings():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol(built):
    lib = ctypes.CDLL(built)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.lib().advh_version()


def test_code_object_is_gfx950(built):
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:               # --offloading writes the extracted bundles next to its input
        copy = shutil.copy(built, tmp)
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy],
                             capture_output=True, text=True, cwd=tmp).stdout
    assert "gfx950" in out


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "xai-audio-deepfakes_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_argument_errors_of_the_round3_entry_points(built):
    """The error contract of include/addvisor_hip.h (negative return, nothing launched) for the entry points added in round 3:
    argument validation happens before any HIP call, so it can be exercised without a GPU."""
    import ctypes as C
    lib = _lib.lib()
    EINVAL, EUNSUPPORTED = -1, -4
    buf = (C.c_float * 64)()
    p = C.addressof(buf)
    assert lib.advh_split_f32(None, p, 64, 64, None) == EINVAL
    assert lib.advh_split_f32(p, p, 8, 64, None) == EINVAL                               # lo plane closer than n elements
    assert lib.advh_split_f32(p, p, 64, 0, None) == EINVAL
    assert lib.advh_attention_bwd_split(None, 8, p, 8, p, 8, 1, 16, 64, 1, None) == EINVAL
    assert lib.advh_attention_bwd_split(p, 8, p, 8, p, 8, 1, 16, 64, 3, None) == EINVAL  # H % heads
    assert lib.advh_attention_bwd_split(p, 8, p, 8, p, 8, 1, 300, 64, 1, None) == EUNSUPPORTED   # T > 256
    assert lib.advh_attention_bwd_split(p, 4, p, 8, p, 8, 1, 16, 64, 1, None) == EINVAL  # plane distance not a multiple of 8
    from addvisor_hip.unet_train import Wgrad2dDesc
    d = Wgrad2dDesc(B=1, H=8, W_=16, PHx=1, PWx=1, PHz=1, PWz=1)
    d.X, d.DZ, d.partial = p, p, p
    assert lib.advh_conv_wgrad2d_split(C.byref(d), 48, 32, 48, 0, 32, 0, 1024, 1024, p, None) == EUNSUPPORTED   # CI not 32 | 64
    assert lib.advh_conv_wgrad2d_split(C.byref(d), 32, 32, 64, 40, 32, 0, 1024, 1024, p, None) == EINVAL        # slice past the map's channels
    assert lib.advh_conv_wgrad2d_split(C.byref(d), 32, 32, 32, 0, 32, 0, 0, 1024, p, None) == EINVAL            # no lo plane
    d.PHx = 0
    assert lib.advh_conv_wgrad2d_split(C.byref(d), 32, 32, 32, 0, 32, 0, 1024, 1024, p, None) == EINVAL         # the patch needs a halo
    assert lib.advh_conv_wgrad2d_split_parts(64, 64, 64, 128, 196) == 256 and lib.advh_conv_wgrad2d_split_parts(32, 32, 1, 16, 16) == 1
    from addvisor_hip.gemm import TapsDesc, taps_split_tile
    td = TapsDesc()
    td.X, td.W, td.out_h, td.M, td.Hg, td.Wg, td.h0, td.h1, td.w0, td.w1, td.ntap = p, p, p, 64, 1, 64, 0, 1, 0, 64, 3
    assert lib.advh_conv_taps_split(C.byref(td), 32, 4096, 4096, 0, 4096, None) == EUNSUPPORTED       # 64 channels only
    assert lib.advh_conv_taps_split(C.byref(td), 64, 0, 4096, 0, 4096, None) == EINVAL                # no lo plane
    td.pre_act = 1
    assert lib.advh_conv_taps_split(C.byref(td), 64, 4096, 4096, 0, 4096, None) == EUNSUPPORTED       # no in-buffer activation in this form
    assert lib.advh_conv_taps_split_tile(64, 11, 50) == taps_split_tile(64, 11, 50) == 256 and lib.advh_conv_taps_split_tile(32, 3, 2) == 0
    assert lib.advh_unet_skip_wgrad_split(p, 1024, 8, 16, 1, 16, 16, p, 1, 1, p, p, None) == EINVAL      # H > Fq: the crop must lie inside the input
    assert lib.advh_unet_skip_wgrad_split(p, 0, 16, 16, 1, 16, 16, p, 1, 1, p, p, None) == EINVAL         # no lo plane
    assert lib.advh_unet_skip_wgrad(None, 16, 16, 1, 16, 16, p, 1, 1, p, p, None) == EINVAL
    assert lib.advh_set_option(b"attention_bwd_mfma_f32", 1) == 0 and lib.advh_set_option(b"attention_bwd_mfma_f32", 0) == 0
    assert lib.advh_set_option(b"no_such_option", 1) == EINVAL
    assert lib.advh_split_overflow(0) == 0                                                # no device initialised: no flag word, reads as clear
