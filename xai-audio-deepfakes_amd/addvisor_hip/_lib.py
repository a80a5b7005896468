"""ctypes binding of ``libaddvisor_hip.so`` (include/addvisor_hip.h).

The library is the product: if it is missing or fails to load this module raises -- there is no
CPU or eager-PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADDVISOR_HIP_LIB") or os.path.join(_HERE, "libaddvisor_hip.so")   # the override is for A/B runs of two builds
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

ERRORS = {-1: "ADVH_EINVAL (bad argument)", -2: "ADVH_ELAUNCH (HIP launch failed)",
          -3: "ADVH_ENOTINIT (advh_init not called)", -4: "ADVH_EUNSUPPORTED"}


class AdvhError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 (``make -C csrc``); hipcc cross-compiles without a GPU."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise AdvhError("building libaddvisor_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes): exactly the symbols include/addvisor_hip.h declares
SIGNATURES = {
    "advh_version": (C.c_char_p, []),
    "advh_init": (_i, []),
    "advh_set_option": (_i, [C.c_char_p, _i]),
    "advh_stft_forward": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "advh_istft_masked": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "advh_istft_masked_c64": (_i, [_p, _p, _i, _i, _i, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "advh_istft_c64": (_i, [_p, _p, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "advh_gemm_f16": (_i, [_p, _i, _p]),
    "advh_w2v2_frontend": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "advh_layernorm": (_i, [_p, _i, _i64, _p, _p, _p, _p, _i64, _i, _i, _f, _i, _p]),
    "advh_layernorm_add": (_i, [_p, _i, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i, _i, _f, _i, _p]),
    "advh_posconv_gather": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "advh_attention_f16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "advh_pool_logreg": (_i, [_p, _p, _f, _p, _p, _p, _i, _i, _i, _p]),
    "advh_unet_stem": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _f, _p]),
    "advh_unet_pack_x": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p]),
    "advh_unet_head": (_i, [_p, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p]),
    "advh_lmac_metrics_accumulate": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "advh_hifigan_pack_mel": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "advh_hifigan_pack_mel_pad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "advh_halo_fill_f16": (_i, [_p, _i, _i, _i, _i, _i, _p]),
    "advh_hifigan_mrf_mix": (_i, [_p, _p, _p, _p, _f, _i64, _p]),
    "advh_hifigan_conv_post": (_i, [_p, _p, _f, _p, _i, _i, _i, _i, _i, _p]),
    "advh_mel_log": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "advh_layernorm_bwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _f, _i, _i, _p]),
    "advh_attention_bwd_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "advh_pool_logreg_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "advh_w2v2_frontend_bwd_group": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "advh_wave_bwd": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p, _p, _i, _f, _p, _i64, _i, _i, _p]),
    "advh_scale_rows": (_i, [_p, _i, _p, _p, _i, _i64, _i, _p]),
    "advh_attr_finalize": (_i, [_p, _p, _p, _i, _i64, _p]),
    "advh_time_mask": (_i, [_p, _p, _p, _p, _p, _i, _i64, _p]),
    "advh_istft_masked_bwd": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "advh_istft_bandswap": (_i, [_p, _p, _i, _i, _i, _p, _i64, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "advh_bn_partial_count": (_i, []),
    "advh_bn_stats": (_i, [_p, _p, _p, _p, _p]),
    "advh_bn_coef": (_i, [_p, _p, _p, _i, _f, _f, _f, _p, _p, _p, _p, _p]),
    "advh_bn_bwd_coef": (_i, [_p, _p, _i, _f, _f, _p, _p, _p, _p]),
    "advh_bn_apply": (_i, [_p, _p, _p, _f, _p, _p]),
    "advh_bn_bwd_sums": (_i, [_p, _p, _i, _p, _p, _f, _p, _p, _p]),
    "advh_bn_bwd_apply": (_i, [_p, _p, _i, _p, _p, _p, _f, _p, _i64, _i64, _i64, _i64, _p]),
    "advh_transpose_gather": (_i, [_p, _p, _p, _p]),
    "advh_split_f32": (_i, [_p, _p, _i64, _i64, _p]),
    "advh_unet_head_bwd": (_i, [_p, _p, _p, _f, _i64, _p, _p, _i, _p]),
    "advh_unet_head_wgrad": (_i, [_p, _p, _i64, _p, _p, _p]),
    "advh_unet_stem_wgrad": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "advh_conv_wgrad2d_parts": (_i, [_i, _i, _i, _i]),
    "advh_conv_wgrad2d_f16": (_i, [_p, _i, _p, _p]),
    "advh_conv_wgrad2d_split_parts": (_i, [_i, _i, _i, _i, _i]),
    "advh_conv_wgrad2d_split": (_i, [_p, _i, _i, _i, _i, _i, _i, _i64, _i64, _p, _p]),
    "advh_resblock_pair_lds_bytes": (_i, [_i, _i, _i]),
    "advh_resblock_pair_f16": (_i, [_p, _i, _p]),
    "advh_conv_taps_tile": (_i, [_i, _i, _i]),
    "advh_conv_taps_lds_bytes": (_i, [_i, _i, _i]),
    "advh_conv_taps_f16": (_i, [_p, _i, _p]),
    "advh_conv_taps_split_tile": (_i, [_i, _i, _i]),
    "advh_conv_taps_split": (_i, [_p, _i, _i64, _i64, _i64, _i64, _p]),
    "advh_conv_taps2d_f16": (_i, [_p, _i, _p]),
    "advh_upconv21_tile_f16": (_i, [_p, _p]),
    "advh_posconv_tile_f16": (_i, [_p, _p]),
    "advh_conv53s21_tile_f16": (_i, [_p, _p]),
    "advh_conv53s21_tile_lds_bytes": (_i, []),
    "advh_posconv_tile_lds_bytes": (_i, [_i, _i]),
    "advh_w2v2_frontend_split": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "advh_layernorm_split": (_i, [_p, _i, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _p]),
    "advh_posconv_gather_split": (_i, [_p, _p, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "advh_attention_split": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _p]),
    "advh_unet_stem_split": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _i64, _i, _i, _f, _p]),
    "advh_unet_pack_x_split": (_i, [_p, _i, _i, _i, _i, _i, _p, _i64, _i, _i, _i, _i, _p]),
    "advh_hifigan_pack_mel_split": (_i, [_p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "advh_hifigan_mrf_mix_split": (_i, [_p, _p, _p, _p, _f, _i64, _i64, _p]),
    "advh_hifigan_conv_post_split": (_i, [_p, _i64, _p, _f, _p, _i, _i, _i, _i, _i, _p]),
    "advh_unet_head_split": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p]),
    "advh_upconv21_tile_lds_bytes": (_i, []),
    "advh_bn_stats_split": (_i, [_p, _i64, _p, _p, _p, _p]),
    "advh_bn_apply_split": (_i, [_p, _i64, _p, _p, _f, _p, _i64, _p]),
    "advh_bn_bwd_sums_split": (_i, [_p, _i64, _p, _i64, _p, _p, _f, _p, _p, _p]),
    "advh_bn_bwd_apply_split": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _f, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advh_unet_head_bwd_split": (_i, [_p, _p, _p, _f, _i64, _p, _p, _i64, _p]),
    "advh_unet_head_wgrad_split": (_i, [_p, _p, _i64, _i64, _p, _p, _p]),
    "advh_unet_stem_wgrad_split": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "advh_unet_skip_wgrad": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "advh_unet_skip_wgrad_split": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "advh_split_overflow": (_i, [_i]),
    "advh_resblock_pair_x3_lds_bytes": (_i, [_i, _i, _i]),
    "advh_resblock_pair_x3": (_i, [_p, _i, _p]),
    "advh_layernorm_bwd_split": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _p, _i, _p, _p, _i64, _p, _p, _i64, _i, _i, _f, _i, _i, _p]),
    "advh_attention_bwd_split": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _p]),
    "advh_pool_logreg_bwd_split": (_i, [_p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "advh_posconv_gather_bwd_split": (_i, [_p, _p, _i64, _i, _i, _i, _i, _i, _i, _p, _i64, _p]),
    "advh_w2v2_frontend_bwd_group_split": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i, _i, _i, _p]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # torch first: it bundles its own libamdhip64; loading ours afterwards makes the dynamic linker
        # reuse that runtime (same SONAME), so kernels and torch's device pointers share one HIP context.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise AdvhError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no fallback path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class SplitRangeError(AdvhError, FloatingPointError):
    """A kernel of the fp32-class mode met a value outside the split format's range (|x| > 65504)."""


def check_overflow(what: str = "") -> None:
    """Raise if the library's sticky range flag is set (and clear it).  The flag lives in host-mapped memory: reading it does
    not synchronise, so it reports kernels that have ALREADY run -- call after a synchronisation point for a definite answer."""
    if _lib is not None and _lib.advh_split_overflow(1):
        raise SplitRangeError((what + ": " if what else "") + "a value left the fp32-class format's range (|x| > 65504 between two "
                              "matrix products): the split planes saturated.  Scale the input / weights, or run precision='f16' "
                              "diagnostics; the reference's fp32 has no such limit (include/addvisor_hip.h, advh_split_overflow)")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise AdvhError(f"{what}: {ERRORS.get(rc, rc)}")
    check_overflow(what)


_inited = set()


def init() -> None:
    """advh_init() on the current device (twiddle tables, LDS limits); once per device of this process."""
    import torch
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    if dev not in _inited:
        check(lib().advh_init(), "advh_init")
        _inited.add(dev)
