"""Host-side planning for the implicit-GEMM kernel (``advh_gemm_f16``, include/addvisor_hip.h).

A *plan* is the filled ``advh_gemm_desc`` plus the device tensors it points to (packed fp16 weights,
fp32 bias, the K-chunk offset table).  Plans are built once per (layer, batch size) and replayed.
All index arithmetic that decides which addresses a kernel touches lives here and is unit-tested on
the CPU (tests/test_gemm_plan.py) by replaying the descriptor in numpy.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

BK = 64
ACT = {"none": 0, "gelu": 1, "leaky": 2}
WIDE_EPILOGUE = os.environ.get("ADDVISOR_GEMM_WIDE", "1") != "0"      # A/B switch for the 16-byte epilogue


def packed_row_channel(rows: int) -> np.ndarray:
    """Output channel carried by packed weight row R in the wide layout (``advh_gemm_desc.wide``)."""
    R = np.arange(rows)
    return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3)
TILE_AUTO, TILE_128x128, TILE_256x64, TILE_256x32 = 0, 1, 2, 3
TILE_256x128_W8, TILE_128x256_W8 = 9, 10
TILE_BN = {TILE_128x128: 128, TILE_256x64: 64, TILE_256x32: 32, TILE_256x128_W8: 128, TILE_128x256_W8: 256}
TILE_NAMES = {TILE_128x128: "128x128", TILE_256x64: "256x64", TILE_256x32: "256x32", TILE_256x128_W8: "256x128w8", TILE_128x256_W8: "128x256w8"}
# device symbol (as rocprofv3 prints it) of the kernels the live profile covers
TILE_KERNELS = {TILE_128x128: "gemm_f16_kernel<128, 128, 2, 2, 4>", TILE_128x256_W8: "gemm_f16_kernel<128, 256, 2, 4, 3>",
                TILE_256x128_W8: "gemm_f16_kernel<256, 128, 4, 2, 3>"}


class GemmDesc(C.Structure):
    _fields_ = [
        ("A0", C.c_void_p), ("A1", C.c_void_p), ("W", C.c_void_p), ("ktab", C.c_void_p),
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("out_h", C.c_void_p), ("out_f", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("Ktot", C.c_int32), ("w_rows", C.c_int32),
        ("Hg", C.c_int32), ("Wg", C.c_int32), ("h0", C.c_int32), ("h1", C.c_int32),
        ("w0", C.c_int32), ("w1", C.c_int32), ("halo_zero", C.c_int32),
        ("a_sB", C.c_int64 * 2), ("a_sH", C.c_int64 * 2), ("a_sW", C.c_int64 * 2),
        ("a_c0", C.c_int64 * 2), ("a_sZ", C.c_int64 * 2),
        ("w_sZ", C.c_int64), ("bias_sZ", C.c_int64),
        ("o_sB", C.c_int64), ("o_sH", C.c_int64), ("o_sW", C.c_int64), ("o_c0", C.c_int64),
        ("o_sNhi", C.c_int64), ("o_sZ", C.c_int64),
        ("n_div", C.c_int32), ("nz", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
        ("resid_f32", C.c_int32), ("ktab_identity", C.c_int32),
        ("out_h2", C.c_void_p), ("slope2", C.c_float), ("ph_r", C.c_int32), ("ph_pad", C.c_int32), ("ph_T", C.c_int32),
        ("out_pre", C.c_void_p), ("dact_src", C.c_void_p), ("wide", C.c_int32), ("w_ld", C.c_int64), ("sc", C.c_int32), ("n_sub", C.c_int32), ("o_sNhh", C.c_int64),
        ("nz_lo", C.c_int32), ("z_inner", C.c_int32), ("a_sZ2", C.c_int64 * 2), ("o_sZ2", C.c_int64), ("plain", C.c_int32), ("plain_out", C.c_int32),
        ("split", C.c_int32), ("a_lo", C.c_int64 * 2), ("w_lo", C.c_int64), ("o_lo", C.c_int64),
    ]


X3_KERNELS = {TILE_128x128: "gemm_x3_kernel<128, 128, 2, 2", TILE_256x64: "gemm_x3_kernel<256, 64, 4, 1", TILE_256x32: "gemm_x3_kernel<256, 32, 4, 1"}


def kernel_name(tile: int, plain: bool, split: bool = False) -> str:
    """Name of the kernel instantiation a launch runs, as rocprofv3 prints it (the 256-thread kernels carry the
    affine-row flag as their last template argument)."""
    if split:
        return X3_KERNELS[tile] + (", true>" if plain else ", false>")
    k = TILE_KERNELS.get(tile, "gemm_f16_kernel")
    if k.startswith("gemm_f16_kernel<"):
        k = k[:-1] + (", true>" if plain and tile == TILE_128x128 else ", false>")
    return k


class _Profile:
    """Optional live timing of the dominant kernel (the 128x128 tile): HIP events recorded on the launch
    stream around each launch; bench.py reads the totals after a synchronise."""

    def __init__(self):
        self.reset(False)

    def reset(self, enabled: bool):
        self.enabled, self.events = enabled, []

    def summary(self, kernel=None):
        """(ms, flops, launches) of one kernel instantiation (the name rocprofv3 prints, see ``kernel_name``); default: the
        one with the largest total time, remembered in ``self.kernel``."""
        per = {}
        for a, b, f, t, shape in self.events:
            r = per.setdefault(kernel_name(t, shape[4], shape[5] if len(shape) > 5 else False), [0.0, 0.0, 0])
            r[0] += a.elapsed_time(b); r[1] += f; r[2] += 1
        if not per:
            return 0.0, 0.0, 0
        if kernel is None:
            kernel = max(per, key=lambda k: per[k][0])
        self.kernel = kernel
        return tuple(per.get(kernel, (0.0, 0.0, 0)))

    def by_shape(self):
        """{(M, N, K, nz, tile name): [ms, flops, launches]} of the recorded launches (tools/bench_shapes.py)."""
        per = {}
        for a, b, f, t, shape in self.events:
            r = per.setdefault(shape[:4] + (TILE_NAMES.get(t, str(t)) + ("+" if shape[4] else "") + ("x3" if len(shape) > 5 and shape[5] else ""),), [0.0, 0.0, 0])
            r[0] += a.elapsed_time(b); r[1] += f; r[2] += 1
        return per


PROFILE = _Profile()


class _Tuner:
    """Per-plan tile selection by measurement.  While ``active`` every ``GemmPlan.run`` times each candidate
    tile on its real operands (results of that pass are garbage for in-place plans: use it only on a
    throw-away warm-up call, e.g. ``ExplainPipeline.tune``) and keeps the fastest."""
    active = False
    log = []


TUNER = _Tuner()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def split_planes(x: torch.Tensor) -> torch.Tensor:
    """Host-side packer of the split format (csrc/device_math.h): ``x`` (any float dtype) -> fp16 ``[2, *x.shape]`` with
    ``x = hi + lo * 2**-11``; computed in fp64, so an fp64 source (BatchNorm-folded / composed weights) keeps ~22 bits."""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0:
        # device tensors (the training path's per-step weight refresh): one HIP launch, no host synchronisation; out-of-range
        # values saturate and raise the sticky flag (SplitRangeError at the next _lib.check) instead of the ValueError below
        src = x.detach().contiguous()
        if src.data_ptr() % 16:                               # a contiguous view at an odd storage offset: the kernel loads float4
            src = src.clone()
        out = torch.empty((2,) + tuple(src.shape), dtype=torch.float16, device=src.device)
        if src.numel():
            _lib.check(_lib.lib().advh_split_f32(src.data_ptr(), out.data_ptr(), src.numel(), src.numel(),
                                                 torch.cuda.current_stream(src.device).cuda_stream), "advh_split_f32")
        return out
    x64 = x.detach().to(torch.float64)
    # range of the format: |x| <= 65504 (hi is an fp16).  Weights / host tensors outside it are a caller error, reported here
    # rather than as saturated planes on the device (csrc/device_math.h split_f32 saturates and raises the sticky flag).
    bad = ~(x64.abs() <= 65504.0)
    if bool(bad.any()):
        raise ValueError(f"split format: {int(bad.sum())} value(s) outside |x| <= 65504 (or NaN); max |x| = {float(x64.abs().nan_to_num(float('inf')).max()):g}")
    hi = x64.to(torch.float16)
    hi = torch.where(x64.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
    lo = ((x64 - hi.to(torch.float64)) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo])


def join_planes(t: torch.Tensor) -> torch.Tensor:
    """fp32 value of a split-format tensor ``[2, ...]``."""
    return t[0].float() + t[1].float() * (1.0 / 2048.0)


W8_RULE = int(os.environ.get("ADDVISOR_GEMM_W8_MIN_M", "0"))
X3_SUPER_COLUMN_BYTES = int(os.environ.get("ADDVISOR_X3_SC_BYTES", "1600000"))
SUPER_COLUMN_BYTES = int(os.environ.get("ADDVISOR_GEMM_SC_BYTES", "1600000"))   # +5-8 % on isolated QKV / FFN1 launches, +0.7 % in the pipeline


def super_columns(N: int, Kp: int, M: int, bn: int = 128, split: bool = False) -> int:
    """Super-column width (in ``bn``-column tiles) of the tile order: keep the weight slice in flight under
    ``SUPER_COLUMN_BYTES`` when the whole weight would not stay resident in an XCD's 4 MiB L2; 0 = off.  A split-format
    weight is two fp16 planes (4 bytes per element)."""
    bpe = 4 if split else 2
    budget = X3_SUPER_COLUMN_BYTES if split else SUPER_COLUMN_BYTES
    if budget <= 0 or N <= bn:
        return 0
    tiles_n = (N + bn - 1) // bn
    if tiles_n * bn * Kp * bpe <= 3 * 1024 * 1024:
        return 0
    sc = min(tiles_n, budget // (bn * Kp * bpe))
    # narrower super-columns re-read the activations too often (deep-K layers): FFN2 in the fp32-class mode (K = 3072, six
    # column tiles of 1.5 MB) ran 301 / 323 / 359 / 362 TFLOP/s with super-columns of 1 / 2 / 3 / all tiles (warm clock)
    return sc if sc >= (3 if split else 4) else 0


PLAIN_ROWS = os.environ.get("ADDVISOR_GEMM_PLAIN", "1") != "0"          # A/B switch for the affine-row loader


def pick_tile(N: int, M: int = 0) -> Tuple[int, int]:
    """(tile id, BN) the AUTO rule of advh_gemm_f16 picks (same rule as csrc/gemm.hip).  The 512-thread 128x256 /
    256x128 tiles win an isolated-GEMM loop by 10-15 % on the 3B-row shapes but lose ~3 % inside the pipeline
    (profiles/history): they stay available to the tuner, the rule stays 128x128."""
    if N <= 32:
        return TILE_256x32, 32
    if N <= 64:
        return TILE_256x64, 64
    return TILE_128x128, 128


@dataclass
class Source:
    """Addressing of one A source, in 16-byte chunks (8 halfs)."""
    sB: int
    sH: int
    sW: int
    c0: int
    sZ: int = 0
    sZ2: int = 0


class GemmPlan:
    """One launch of the implicit GEMM.  ``ktab_host``/``desc`` stay inspectable for the CPU tests."""

    def __init__(self, *, M: int, N: int, w2: torch.Tensor, ktab: np.ndarray, sources: Sequence[Source],
                 Hg: int, Wg: int, window: Tuple[int, int, int, int], halo_zero: bool,
                 out: Tuple[int, int, int, int], n_div: Optional[int] = None, o_sNhi: int = 0,
                 o_sZ: int = 0, nz: int = 1, bias: Optional[torch.Tensor] = None, act: str = "none",
                 slope: float = 0.0, device=None, w_sZ: Optional[int] = None, bias_sZ: int = 0,
                 slope2: float = 0.0, phase: Tuple[int, int, int] = (0, 0, 0), cache: Optional[tuple] = None,
                 n_sub: int = 0, o_sNhh: int = 0, nz_lo: int = 0, z_inner: bool = False, o_sZ2: int = 0, plain: bool = False,
                 split: bool = False):
        """``w2``: fp32 ``[nz, N, K]`` (K = 8 * len(ktab) before padding) or a zero-argument callable returning it
        (only called when the packed weight is not in ``cache``); ``ktab``: int64 chunk offsets with bit 31 as
        source selector; ``out`` = (o_sB, o_sH, o_sW, o_c0) in elements.  ``cache = (dict, key)`` shares the
        packed fp16 weight / bias device tensors between plans of different batch shapes (the weight of a layer
        does not depend on the batch).  ``split``: fp32-class launch (``desc.split``): the weight is packed as two
        fp16 planes ``[2, nz, w_rows, Kp]`` from its fp32 / fp64 source, activations and fp16 outputs are ``[2, ...]``
        plane pairs (``split_planes`` / ``join_planes``)."""
        K = 8 * len(ktab)
        self.split = bool(split)
        Kp = round_up(K, BK)
        n_div_v = n_div if n_div is not None else round_up(N, 4)
        # 16-byte epilogue stores need 8 consecutive channels per lane (permuted weight rows) and 8-aligned addressing
        wide = WIDE_EPILOGUE and N % 8 == 0 and n_div_v % 8 == 0 and all(int(x) % 8 == 0 for x in (*out, o_sNhi, o_sZ, o_sNhh, o_sZ2))
        tile, BN = pick_tile(N, M)
        if W8_RULE and N % 256 == 0 and M >= W8_RULE and bool((np.asarray(ktab) == np.arange(len(ktab))).all()):
            tile, BN = TILE_128x256_W8, 256                    # experiment switch: plain wide GEMMs on the 512-thread tile
        w_rows = round_up(N, 256)        # any tile's BN divides 256: the tile can be re-chosen later (autotune)
        store, ckey = cache if cache is not None else (None, None)
        hit = store.get(ckey) if store is not None else None
        if hit is None:
            w2t = w2() if callable(w2) else w2
            assert w2t.dim() == 3 and w2t.shape[0] == nz and w2t.shape[1] == N and w2t.shape[2] == K, (w2t.shape, nz, N, K)
            wp = torch.zeros(((2, nz, w_rows, Kp) if split else (nz, w_rows, Kp)), dtype=torch.float16)
            w16 = split_planes(w2t) if split else w2t.to(torch.float16)        # [2, nz, N, K] | [nz, N, K]
            if wide:
                src = packed_row_channel(w_rows)
                keep = src < N
                wp[..., torch.from_numpy(np.nonzero(keep)[0]), :K] = w16[..., torch.from_numpy(src[keep]), :]
            else:
                wp[..., :N, :K] = w16
            wd = wp.to(device) if device is not None else wp
            bd = None
            if bias is not None:
                assert bias.numel() == nz * N or bias_sZ == 0
                bd = bias.to(torch.float32).contiguous()
                bd = bd.to(device) if device is not None else bd
            hit = (wd, bd, wide)
            if store is not None:
                store[ckey] = hit
        assert hit[2] == wide, "a cached packed weight is shared between plans of different output alignment"
        kt = np.concatenate([ktab, np.full((Kp - K) // 8, ktab[0], dtype=np.int64)]).astype(np.int64)
        assert (kt & 0x7FFFFFFF).max() < 2 ** 31
        self.ktab_host = kt
        self.K, self.Kp, self.tile, self.BN = K, Kp, tile, BN
        self.device = device
        self.w, self.bias = hit[0], hit[1]
        kt32 = torch.from_numpy(kt.astype(np.uint32).view(np.int32).copy())
        self.ktab = kt32.to(device) if device is not None else kt32
        d = GemmDesc()
        d.M, d.N, d.Ktot, d.w_rows = M, N, Kp, w_rows
        d.Hg, d.Wg = Hg, Wg
        d.h0, d.h1, d.w0, d.w1 = window
        d.halo_zero = int(halo_zero)
        for s, src in enumerate(sources):
            d.a_sB[s], d.a_sH[s], d.a_sW[s], d.a_c0[s], d.a_sZ[s] = src.sB, src.sH, src.sW, src.c0, src.sZ
            d.a_sZ2[s] = src.sZ2
        d.w_sZ = w_rows * Kp if w_sZ is None else w_sZ
        d.split = int(split)
        d.w_lo = nz * w_rows * Kp if split else 0
        assert not split or w_sZ is None
        d.bias_sZ = bias_sZ
        d.o_sB, d.o_sH, d.o_sW, d.o_c0 = out
        d.o_sNhi, d.o_sZ = o_sNhi, o_sZ
        d.n_sub, d.o_sNhh = n_sub, o_sNhh
        d.nz_lo, d.z_inner, d.o_sZ2 = nz_lo, int(z_inner), o_sZ2
        assert nz_lo <= 1 or nz % nz_lo == 0
        d.n_div = n_div_v
        d.wide = int(wide)
        d.nz = nz
        d.act, d.slope = ACT[act], slope
        d.slope2 = slope2
        d.ph_r, d.ph_pad, d.ph_T = phase
        d.ktab_identity = int(bool((kt == np.arange(len(kt))).all()))
        # affine-row loader of the 128x128 tile: row m at a_c0 + m * a_sW, also for rows the window excludes
        d.plain = int(bool(plain and PLAIN_ROWS and d.ktab_identity and len(sources) == 1))
        if d.plain:
            src = sources[0]
            assert Hg == 1 or src.sH == Wg * src.sW, "plain rows need h * a_sH == (h * Wg) * a_sW"
            assert M <= Hg * Wg or src.sB == Hg * Wg * src.sW, "plain rows need b * a_sB == (b * Hg * Wg) * a_sW"
        o_sB, o_sH, o_sW, _ = out
        d.plain_out = int(bool(d.plain and n_div_v >= N and phase[0] == 0 and n_sub <= 1 and tuple(window) == (0, Hg, 0, Wg)
                               and (Hg == 1 or o_sH == Wg * o_sW) and (M <= Hg * Wg or o_sB == Hg * Wg * o_sW)))
        d.sc = super_columns(N, Kp, M, 128, split)
        self.desc = d
        self.nsrc = len(sources)
        self.flops = 2.0 * M * N * K * nz          # algorithmic (unpadded) FLOPs of this launch

    def load_weights(self, w2: torch.Tensor, bias: Optional[torch.Tensor] = None):
        """Re-pack ``w2 [nz, N, K]`` (a tensor on the plan's device, any float dtype) into the plan's fp16 operand in
        place (both planes of the split format for an fp32-class plan), with the wide-epilogue row permutation if the plan
        uses it -- the per-step weight refresh of the training path (no host round trip; a handful of torch copy kernels)."""
        N, K = self.desc.N, self.K
        assert w2.shape == (self.desc.nz, N, K) and w2.device == self.w.device, (w2.shape, (self.desc.nz, N, K))
        w16 = split_planes(w2) if self.split else w2.to(torch.float16)           # [2, nz, N, K] | [nz, N, K]
        if self.desc.wide:
            if not hasattr(self, "_perm"):
                src = packed_row_channel(self.w.shape[-2])
                keep = np.nonzero(src < N)[0]
                self._perm = (torch.from_numpy(keep).to(self.w.device), torch.from_numpy(src[keep]).to(self.w.device))
            dst_rows, src_rows = self._perm
            self.w[..., dst_rows, :K] = w16[..., src_rows, :]
        else:
            self.w[..., :N, :K] = w16
        if bias is not None:
            self.bias.copy_(bias.reshape(self.bias.shape).to(torch.float32))

    def run(self, A0: torch.Tensor, A1: Optional[torch.Tensor] = None, *, out_h: Optional[torch.Tensor] = None,
            out_f: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, stream: Optional[int] = None,
            out_h2: Optional[torch.Tensor] = None, out_pre: Optional[torch.Tensor] = None,
            dact_src: Optional[torch.Tensor] = None):
        d = self.desc
        d.out_h2 = out_h2.data_ptr() if out_h2 is not None else None
        d.out_pre = out_pre.data_ptr() if out_pre is not None else None
        d.dact_src = dact_src.data_ptr() if dact_src is not None else None
        assert out_pre is None or out_pre.dtype == torch.float16
        assert dact_src is None or dact_src.dtype == torch.float16
        assert A0.dtype == torch.float16 and A0.is_cuda
        if self.split:                                     # plane pairs [2, ...]: lo plane = stride(0) elements behind
            o_lo = None
            for s_, t in enumerate((A0, A1)):
                if t is not None:
                    assert t.shape[0] == 2 and t.stride(0) % 8 == 0, "split operand must be a [2, ...] plane pair"
                    d.a_lo[s_] = t.stride(0) // 8
            for t in (out_h, out_h2, out_pre, dact_src, resid if (resid is not None and resid.dtype == torch.float16) else None):
                if t is not None:
                    assert t.shape[0] == 2 and t.stride(0) % 8 == 0 and o_lo in (None, t.stride(0)), "split outputs share one plane pitch"
                    o_lo = t.stride(0)
            d.o_lo = o_lo or 0
        d.A0 = A0.data_ptr()
        d.A1 = A1.data_ptr() if A1 is not None else None
        assert (A1 is not None) == (self.nsrc == 2)
        if d.plain:
            # affine-row loader: row m reads Ktot contiguous halfs at chunk a_c0 + m * a_sW for EVERY m < M (filler rows too):
            # the furthest byte must lie inside the operand (its plane, in split mode) -- the slack rows behind the
            # feature-encoder buffers exist for this; a mismatch between the allocation and the plan must fail here, on
            # the host, not as an out-of-bounds DMA on the device
            plane = A0.stride(0) if self.split else A0.numel()
            need = (int(d.a_c0[0]) + (d.M - 1) * int(d.a_sW[0])) * 8 + d.Ktot + int(d.a_sZ[0]) * 8 * max(0, d.nz - 1)
            if need > plane:
                raise ValueError(f"affine-row GEMM operand too small: the plan reads {need} halfs per plane, the tensor holds {plane} "
                                 "(missing slack rows behind the buffer?)")
        d.W, d.ktab = self.w.data_ptr(), self.ktab.data_ptr()
        d.bias = self.bias.data_ptr() if self.bias is not None else None
        d.out_h = out_h.data_ptr() if out_h is not None else None
        d.out_f = out_f.data_ptr() if out_f is not None else None
        if out_h is not None:
            assert out_h.dtype == torch.float16
        if out_f is not None:
            assert out_f.dtype == torch.float32
        if resid is not None:
            assert resid.dtype in (torch.float16, torch.float32)
            d.resid, d.resid_f32 = resid.data_ptr(), int(resid.dtype == torch.float32)
        else:
            d.resid, d.resid_f32 = None, 0
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        if self.split:                                     # the super-column width counts tiles of the chosen width
            d.sc = super_columns(d.N, self.Kp, d.M, TILE_BN.get(self.tile, 128), True)
        if TUNER.active and not getattr(self, "_tuned", False):
            self._tune(d, stream)
        prof = PROFILE.enabled and self.tile not in (TILE_256x64, TILE_256x32)
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(_lib.lib().advh_gemm_f16(C.byref(d), self.tile, stream), "advh_gemm_f16")
        if prof:
            e1.record()
            PROFILE.events.append((e0, e1, self.flops, self.tile, (self.desc.M, self.desc.N, self.K, self.desc.nz, bool(self.desc.plain), self.split)))


def _tune(self, d, stream):
    cands = [TILE_128x128, TILE_128x256_W8, TILE_256x128_W8] if (self.desc.N > 64 and not self.split) else [self.tile]
    best, best_ms = self.tile, None
    if len(cands) > 1:
        for t in cands:
            _lib.check(_lib.lib().advh_gemm_f16(C.byref(d), t, stream), "advh_gemm_f16")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _lib.check(_lib.lib().advh_gemm_f16(C.byref(d), t, stream), "advh_gemm_f16")
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 3
            if best_ms is None or ms < best_ms:
                best, best_ms = t, ms
        TUNER.log.append((self.desc.M, self.desc.N, self.K, self.desc.nz, TILE_NAMES[best], best_ms))
    self.tile, self._tuned = best, True


GemmPlan._tune = _tune


# ------------------------------------------------------------------------------------------ layouts
@dataclass
class FMap:
    """Zero-haloed NHWC fp16 feature map ``[B, H+2PH, W+2PW, C]`` (C % 8 == 0)."""
    B: int
    H: int
    W: int
    C: int
    PH: int
    PW: int
    t: Optional[torch.Tensor] = None
    split: bool = False          # fp32-class mode: ``t`` is the plane pair [2, B, Hp, Wp, C]

    @property
    def Hp(self):
        return self.H + 2 * self.PH

    @property
    def Wp(self):
        return self.W + 2 * self.PW

    def alloc(self, device):
        shape = (self.B, self.Hp, self.Wp, self.C)
        self.t = torch.zeros(((2,) + shape) if self.split else shape, dtype=torch.float16, device=device)
        return self

    def interior(self) -> torch.Tensor:
        """Interior view (of the hi plane in split mode)."""
        t = self.t[0] if self.split else self.t
        return t[:, self.PH:self.PH + self.H, self.PW:self.PW + self.W, :]


def plan_linear(M: int, weight: torch.Tensor, bias: Optional[torch.Tensor], *, lda: Optional[int] = None,
                ldo: Optional[int] = None, o_c0: int = 0, act: str = "none", device=None, cache=None,
                split: bool = False) -> GemmPlan:
    """``out[m, :N] = act(A[m, :K] @ weight.T + bias)``; nn.Linear (modeling_wav2vec2.py:422-572)."""
    N, K = weight.shape
    assert K % 8 == 0
    lda = K if lda is None else lda
    ldo = N if ldo is None else ldo
    assert lda % 8 == 0 and ldo % 4 == 0
    return GemmPlan(M=M, N=N, w2=lambda: weight[None].float(), ktab=np.arange(K // 8, dtype=np.int64),
                    sources=[Source(0, 0, lda // 8, 0)], Hg=1, Wg=M, window=(0, 1, 0, M), halo_zero=False,
                    out=(0, 0, ldo, o_c0), bias=bias, act=act, device=device, cache=cache, plain=True, split=split)


def plan_conv1d_cl(B: int, P_in: int, P_out: int, L_out: int, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   stride: int, *, act: str = "gelu", compact_out: bool = False, device=None, cache=None,
                   slack_rows: int = 0, split: bool = False) -> GemmPlan:
    """Channels-last Conv1d (no padding) as an overlapping-row GEMM: wav2vec2 feature-encoder layers 1-6
    (modeling_wav2vec2.py:254-323).  Input ``[B, P_in, Cin]``, rows >= L_in are zero filler; output
    ``[B, P_out, Cout]`` with rows >= L_out written as zeros, or ``[B, L_out, Cout]`` if ``compact_out``.
    ``slack_rows``: readable rows the caller allocated behind the input; with ``>= k - stride`` (the reach of the last
    filler row) and ``P_in == P_out * stride`` the launch uses the affine-row loader (``desc.plain``)."""
    Cout, Cin, k = weight.shape
    assert Cin % 8 == 0
    w2 = lambda: weight.permute(0, 2, 1).reshape(1, Cout, k * Cin).float()  # K order: (tap, channel)
    out = (L_out * Cout, 0, Cout, 0) if compact_out else (P_out * Cout, 0, Cout, 0)
    return GemmPlan(M=B * P_out, N=Cout, w2=w2, ktab=np.arange(k * Cin // 8, dtype=np.int64),
                    sources=[Source(P_in * Cin // 8, 0, stride * Cin // 8, 0)], Hg=1, Wg=P_out,
                    window=(0, 1, 0, L_out), halo_zero=not compact_out, out=out, bias=bias, act=act, device=device,
                    cache=cache, plain=(slack_rows >= k - stride and P_in == P_out * stride), split=split)


def plan_conv2d(srcs: Sequence[FMap], dst: FMap, weight: torch.Tensor, bias: Optional[torch.Tensor], *,
                stride=(1, 1), padding=(1, 1), dilation=(1, 1), act: str = "leaky", slope: float = 0.2,
                dst_c0: int = 0, device=None) -> GemmPlan:
    """nn.Conv2d on zero-haloed NHWC maps (addvisor.py:12-60); ``srcs`` are concatenated along channels
    by pointer (torch.cat of addvisor.py:70-80).  Enumerates the padded OUTPUT grid and writes its halo
    as zeros, so ``dst`` is complete after one launch."""
    Cout, Cin, KH, KW = weight.shape
    assert sum(s.C for s in srcs) == Cin and len(srcs) <= 2
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    B = dst.B
    Ho = (srcs[0].H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    Wo = (srcs[0].W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    assert (Ho, Wo) == (dst.H, dst.W), ((Ho, Wo), (dst.H, dst.W))
    ktabs, ws, sources = [], [], []
    c_lo = 0
    for s, f in enumerate(srcs):
        assert f.C % 8 == 0 and f.PH >= ph and f.PW >= pw and (f.H, f.W, f.B) == (srcs[0].H, srcs[0].W, B)
        cc = f.C // 8
        taps = (np.arange(KH)[:, None] * dh * f.Wp + np.arange(KW)[None, :] * dw).reshape(-1)   # positions
        kt = (taps[:, None] * cc + np.arange(cc)[None, :]).reshape(-1).astype(np.int64) | (s << 31)
        ktabs.append(kt)
        ws.append(weight[:, c_lo:c_lo + f.C].permute(0, 2, 3, 1).reshape(Cout, KH * KW * f.C))
        c0 = ((-sh * dst.PH + f.PH - ph) * f.Wp + (-sw * dst.PW + f.PW - pw)) * cc
        sources.append(Source(f.Hp * f.Wp * cc, sh * f.Wp * cc, sw * cc, c0))
        c_lo += f.C
    Ct = dst.C
    split = dst.split
    assert all(f.split == split for f in srcs)
    return GemmPlan(M=B * dst.Hp * dst.Wp, N=Cout, w2=torch.cat(ws, 1)[None] if split else torch.cat(ws, 1)[None].float(), ktab=np.concatenate(ktabs), split=split,
                    sources=sources, Hg=dst.Hp, Wg=dst.Wp,
                    window=(dst.PH, dst.PH + dst.H, dst.PW, dst.PW + dst.W), halo_zero=True,
                    out=(dst.Hp * dst.Wp * Ct, dst.Wp * Ct, Ct, dst_c0), bias=bias, act=act, slope=slope,
                    device=device)


def plan_convT2d(src: FMap, dst: FMap, weight: torch.Tensor, bias: torch.Tensor, *, stride, dst_c0: int = 0,
                 device=None) -> GemmPlan:
    """nn.ConvTranspose2d with kernel == stride (addvisor.py:45-54): a GEMM over the input pixels whose epilogue
    scatters all kh*kw sub-pixels (column block q = i*sw + j -> row offset i, column offset j: the two-level column
    split of the descriptor), so the input is read once.  Only ``dst``'s interior is written; its halo must already be
    zero."""
    Cin, Cout, KH, KW = weight.shape
    sh, sw = stride
    assert (KH, KW) == (sh, sw) and src.C == Cin and Cin % 8 == 0 and Cout % 4 == 0
    assert (dst.H, dst.W) == (src.H * sh, src.W * sw)
    cc = Cin // 8
    Ct = dst.C
    # w2[n = (i*sw + j)*Cout + co][ci]
    w2 = weight.permute(2, 3, 1, 0).reshape(1, sh * sw * Cout, Cin)
    w2 = w2 if dst.split else w2.float()
    assert src.split == dst.split
    b2 = bias.float().repeat(sh * sw)
    o_c0 = ((-sh * src.PH + dst.PH) * dst.Wp + (-sw * src.PW + dst.PW)) * Ct + dst_c0
    return GemmPlan(M=src.B * src.Hp * src.Wp, N=sh * sw * Cout, w2=w2, ktab=np.arange(cc, dtype=np.int64),
                    sources=[Source(src.Hp * src.Wp * cc, src.Wp * cc, cc, 0)], Hg=src.Hp, Wg=src.Wp,
                    window=(src.PH, src.PH + src.H, src.PW, src.PW + src.W), halo_zero=False,
                    out=(dst.Hp * dst.Wp * Ct, sh * dst.Wp * Ct, sw * Ct, o_c0), n_div=Cout,
                    o_sNhi=Ct if sw > 1 else dst.Wp * Ct, n_sub=sw if sw > 1 else 0, o_sNhh=dst.Wp * Ct if sw > 1 else 0,
                    bias=b2, bias_sZ=0, act="none", device=device, split=dst.split)


class PlanGroup:
    """Several launches that together produce one map (``plan_upconv2d``); same ``run`` signature as one plan."""

    def __init__(self, plans: Sequence[GemmPlan], flops: float):
        self.plans, self.flops = list(plans), flops
        self.tile = self.plans[0].tile

    def run(self, A0: torch.Tensor, A1: Optional[torch.Tensor] = None, *, out_h: torch.Tensor, stream: Optional[int] = None):
        for p in self.plans:
            p.run(A0, A1, out_h=out_h, stream=stream)


def add_indicator(f: FMap, channel: int) -> FMap:
    """Set channel ``channel`` of an allocated map to 1 inside the image (halo stays 0): the in-image indicator the
    fused up-convolution multiplies the transposed convolution's bias with."""
    f.interior()[..., channel] = 1.0          # split mode: 1.0 = (hi 1, lo 0)
    return f


def compose_upconv_weights(wt: torch.Tensor, bt: torch.Tensor, wc: torch.Tensor, stride, indicator: Tuple[str, int]):
    """Composed fp64 weights of ``conv3x3(cat([ConvTranspose2d(coarse), skip]))`` per output parity class (see
    ``plan_upconv2d``): returns ``([N, K] per class z = ph * sw + pw, cu0, cu1)`` with K = (coarse taps x 8 cu0 channels,
    then the 9 fine taps x 8 cu1 channels); ``cu0`` / ``cu1`` = 16-byte chunks read per coarse / fine tap."""
    sh, sw = stride
    Cb, Cu = wt.shape[:2]
    N, Cs = wc.shape[0], wc.shape[1] - Cu
    where, ich = indicator
    assert where in ("coarse", "skip") and ich >= (Cb if where == "coarse" else Cs)
    wt64, wc64, bt64 = wt.double(), wc.double(), bt.double()
    nth, ntw = 2, (2 if sw == 2 else 3)
    cu0 = (max(Cb, ich + 1 if where == "coarse" else 0) + 7) // 8
    cu1 = (max(Cs, ich + 1 if where == "skip" else 0) + 7) // 8
    w1 = torch.zeros(N, 3, 3, cu1 * 8, dtype=torch.float64)           # skip part: plain 3x3 taps, the same for every parity
    w1[..., :Cs] = wc64[:, Cu:].permute(0, 2, 3, 1)
    if where == "skip":
        w1[..., ich] = torch.einsum("nukl,u->nkl", wc64[:, :Cu], bt64)
    w2 = []
    for ph in range(sh):
        dh_min = (ph - 1) // sh
        for pw in range(sw):
            dw_min = (pw - 1) // sw
            w0 = torch.zeros(N, nth, ntw, cu0 * 8, dtype=torch.float64)
            for kh in range(3):
                for kw in range(3):
                    r, c = ph + kh - 1, pw + kw - 1
                    ti, tj = r // sh - dh_min, c // sw - dw_min
                    w0[:, ti, tj, :Cb] += torch.einsum("nu,bu->nb", wc64[:, :Cu, kh, kw], wt64[:, :, r % sh, c % sw])
                    if where == "coarse":
                        w0[:, ti, tj, ich] += wc64[:, :Cu, kh, kw] @ bt64
            w2.append(torch.cat([w0.reshape(N, -1), w1.reshape(N, -1)], 1))
    return w2, cu0, cu1


def plan_upconv2d(coarse: FMap, skip: FMap, dst: FMap, wt: torch.Tensor, bt: torch.Tensor, wc: torch.Tensor,
                  bc: torch.Tensor, *, stride, coarse_C: int, skip_C: int, indicator: Tuple[str, int],
                  slope: float = 0.2, device=None) -> PlanGroup:
    """``conv3x3(cat([ConvTranspose2d(coarse), skip])) + bias -> LeakyReLU`` (addvisor.py:45-46,69-71 and the three
    blocks after it) WITHOUT materialising the up-sampled map.  A transposed convolution with kernel == stride is
    pointwise in its input pixel, so for output pixels of one parity class (oh % sh, ow % sw) the 3x3 window over the
    up-sampled map is a small convolution over the COARSE map with composed weights
    ``sum_cu wc[n, cu, kh, kw] * wt[cb, cu, (ph+kh-1) % sh, (pw+kw-1) % sw]`` at coarse tap
    ``((ph+kh-1) // sh, (pw+kw-1) // sw)``: 2 coarse taps per strided axis, 3 per unstrided one.  The transposed
    convolution's bias reaches an output only through window positions inside the image, so it is multiplied with an
    in-image indicator channel (``indicator = ("coarse" | "skip", channel)``, set once by ``add_indicator``; zero in the halo)
    through the composed weight ``sum_cu wc[n, cu, kh, kw] * bt[cu]``.  One launch per column parity, row parities in grid z.

    ``wt [Cb, Cu, sh, sw]``, ``bt [Cu]``: the ConvTranspose2d; ``wc [N, Cu + Cs, 3, 3]``, ``bc [N]``: the (BN-folded)
    convolution; ``coarse_C`` / ``skip_C``: real channel counts (the maps may be wider: indicator chunk, zero padding).
    Only ``dst``'s interior is written; its halo must already be zero."""
    sh, sw = stride
    Cb, Cu = wt.shape[:2]
    N = wc.shape[0]
    Cs = wc.shape[1] - Cu
    assert (wt.shape[2], wt.shape[3]) == (sh, sw) and sh == 2 and sw in (1, 2) and wc.shape[2:] == (3, 3)
    assert coarse_C == Cb and skip_C == Cs and coarse.C >= Cb and skip.C >= Cs and coarse.C % 8 == 0 and skip.C % 8 == 0
    assert (dst.H, dst.W) == (coarse.H * sh, coarse.W * sw) == (skip.H, skip.W) and dst.B == coarse.B == skip.B
    assert coarse.PH >= 1 and coarse.PW >= 1 and skip.PH >= 1 and skip.PW >= 1 and dst.C % 8 == 0
    nth, ntw = 2, (2 if sw == 2 else 3)
    cc0, cc1, Ct = coarse.C // 8, skip.C // 8, dst.C                       # pixel pitch of the maps in chunks
    w2, cu0, cu1 = compose_upconv_weights(wt, bt, wc, stride, indicator)    # cu: chunks per tap actually read (<= pitch:
    assert cu0 <= cc0 and cu1 <= cc1                                        # a map may be padded to a 128-byte pitch)
    where = indicator[0]
    kt1 = ((np.arange(3)[:, None] * skip.Wp + np.arange(3)[None, :]).reshape(-1)[:, None] * cc1
           + np.arange(cu1)[None, :]).reshape(-1).astype(np.int64) | (1 << 31)
    kt0 = ((np.arange(nth)[:, None] * coarse.Wp + np.arange(ntw)[None, :]).reshape(-1)[:, None] * cc0
           + np.arange(cu0)[None, :]).reshape(-1).astype(np.int64)
    # rows = coarse pixels (h, w); batch z = ph * sw + pw.  First coarse tap of parity (0, 0) is (h - 1, w - 1); a row
    # parity moves every operand down one (coarse / fine) row, a column parity one pixel to the right.
    c0_0 = ((coarse.PH - 1) * coarse.Wp + coarse.PW - 1) * cc0
    c0_1 = ((skip.PH - 1) * skip.Wp + skip.PW - 1) * cc1
    o_c0 = (dst.PH * dst.Wp + dst.PW) * Ct
    plan = GemmPlan(
        M=coarse.B * coarse.H * coarse.W, N=N, w2=torch.stack(w2) if dst.split else torch.stack(w2).float(), ktab=np.concatenate([kt0, kt1]), split=dst.split,
        sources=[Source(coarse.Hp * coarse.Wp * cc0, coarse.Wp * cc0, cc0, c0_0, sZ=coarse.Wp * cc0, sZ2=cc0 if sw == 2 else 0),
                 Source(skip.Hp * skip.Wp * cc1, sh * skip.Wp * cc1, sw * cc1, c0_1, sZ=skip.Wp * cc1, sZ2=cc1 if sw == 2 else 0)],
        Hg=coarse.H, Wg=coarse.W, window=(0, coarse.H, 0, coarse.W), halo_zero=False,
        out=(dst.Hp * dst.Wp * Ct, sh * dst.Wp * Ct, sw * Ct, o_c0), o_sZ=dst.Wp * Ct, o_sZ2=Ct if sw == 2 else 0,
        nz=sh * sw, nz_lo=sw, z_inner=os.environ.get("ADDVISOR_UPCONV_Z_INNER", "1") != "0", bias=bc, bias_sZ=0, act="leaky", slope=slope, device=device)
    plans = [plan]
    flops = 2.0 * dst.B * dst.H * dst.W * N * (nth * ntw * Cb + 9 * Cs + (nth * ntw if where == "coarse" else 9))
    return PlanGroup(plans, flops)


class ConvS21Desc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out_h", C.c_void_p), ("B", C.c_int),
                ("Ho", C.c_int), ("W_", C.c_int), ("PHi", C.c_int), ("PWi", C.c_int), ("PHo", C.c_int), ("PWo", C.c_int),
                ("act", C.c_int), ("slope", C.c_float)]


def conv_s21_supported(srcs: Sequence[FMap], dst: FMap, weight: torch.Tensor, stride=(1, 1), padding=(1, 1),
                       dilation=(1, 1)) -> bool:
    """Geometry of ``advh_conv53s21_tile_f16``: e2.block.0 of the U-Net (32 -> 64 channels, 5 x 3, stride (2, 1), padding (2, 1))."""
    return (len(srcs) == 1 and tuple(weight.shape) == (64, 32, 5, 3) and tuple(stride) == (2, 1) and tuple(padding) == (2, 1)
            and tuple(dilation) == (1, 1) and srcs[0].C == 32 and dst.C == 64 and srcs[0].PH >= 2 and srcs[0].PW >= 1
            and srcs[0].H == 2 * dst.H and srcs[0].W == dst.W)


class ConvS21TilePlan:
    """e2.block.0 as one LDS line-tile launch (csrc/conv_s21_tile.hip); same ``run`` signature as a GemmPlan."""

    def __init__(self, src: FMap, dst: FMap, weight: torch.Tensor, bias: torch.Tensor, *, slope: float = 0.2, device=None):
        assert conv_s21_supported([src], dst, weight, (2, 1), (2, 1))
        R = np.arange(64)
        ch = 32 * (R >> 5) + 8 * ((R >> 2) & 3) + 4 * ((R >> 4) & 1) + (R & 3)      # MFMA row R carries this output channel
        wp = weight.float()[torch.from_numpy(ch)].permute(2, 3, 0, 1).reshape(15, 64, 32).to(torch.float16).contiguous()
        self.w = wp.to(device) if device is not None else wp
        self.bias = bias.to(torch.float32).contiguous()
        self.bias = self.bias.to(device) if device is not None else self.bias
        d = ConvS21Desc()
        d.B, d.Ho, d.W_ = dst.B, dst.H, dst.W
        d.PHi, d.PWi, d.PHo, d.PWo = src.PH, src.PW, dst.PH, dst.PW
        d.act, d.slope = ACT["leaky"], slope
        self.desc = d
        self.numels = (src.B * src.Hp * src.Wp * 32, dst.B * dst.Hp * dst.Wp * 64)
        self.flops = 2.0 * dst.B * dst.H * dst.W * 64 * 32 * 15
        self.tile = None

    def run(self, A0: torch.Tensor, A1=None, *, out_h: torch.Tensor, stream: Optional[int] = None):
        d = self.desc
        for t, n in zip((A0, out_h), self.numels):
            assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous() and t.numel() == n
        d.X, d.W, d.bias, d.out_h = A0.data_ptr(), self.w.data_ptr(), self.bias.data_ptr(), out_h.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().advh_conv53s21_tile_f16(C.byref(d), stream), "advh_conv53s21_tile_f16")


class UpconvDesc(C.Structure):
    _fields_ = [("Xc", C.c_void_p), ("Xs", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out_h", C.c_void_p),
                ("B", C.c_int), ("Hc", C.c_int), ("W_", C.c_int), ("PHc", C.c_int), ("PWc", C.c_int), ("PHs", C.c_int),
                ("PWs", C.c_int), ("PHo", C.c_int), ("PWo", C.c_int), ("act", C.c_int), ("slope", C.c_float)]


def upconv_tile_supported(coarse: FMap, skip: FMap, dst: FMap, wt: torch.Tensor, wc: torch.Tensor, stride, indicator) -> bool:
    """Geometry of ``advh_upconv21_tile_f16``: up1 + d1.block.0 of the U-Net (64 coarse channels, 8-channel skip map
    carrying the indicator in channel 1, 32 outputs, stride (2, 1))."""
    return (tuple(stride) == (2, 1) and tuple(wt.shape) == (64, 32, 2, 1) and tuple(wc.shape) == (32, 33, 3, 3)
            and coarse.C == 64 and skip.C == 8 and dst.C == 32 and tuple(indicator) == ("skip", 1) and coarse.H % 8 == 0
            and min(coarse.PH, coarse.PW, skip.PH, skip.PW) >= 1)


class UpconvTilePlan:
    """up1 + d1.block.0 as one LDS line-tile launch (csrc/upconv_tile.hip); same ``run`` signature as a GemmPlan."""

    def __init__(self, coarse: FMap, skip: FMap, dst: FMap, wt: torch.Tensor, bt: torch.Tensor, wc: torch.Tensor,
                 bc: torch.Tensor, *, slope: float = 0.2, device=None):
        assert upconv_tile_supported(coarse, skip, dst, wt, wc, (2, 1), ("skip", 1))
        w2, cu0, cu1 = compose_upconv_weights(wt, bt, wc, (2, 1), ("skip", 1))
        assert (cu0, cu1) == (8, 1) and w2[0].shape == (32, 456)
        R = np.arange(32)
        ch = 8 * ((R >> 2) & 3) + 4 * ((R >> 4) & 1) + (R & 3)           # MFMA row R carries this output channel
        wp = torch.zeros(2, 15, 32, 32, dtype=torch.float16)
        for ph in range(2):
            full = torch.zeros(32, 480, dtype=torch.float64)
            full[:, :456] = w2[ph]
            wp[ph] = full[torch.from_numpy(ch)].reshape(32, 15, 32).permute(1, 0, 2).to(torch.float16)
        self.w = wp.contiguous().to(device) if device is not None else wp.contiguous()
        self.bias = bc.to(torch.float32).contiguous()
        self.bias = self.bias.to(device) if device is not None else self.bias
        d = UpconvDesc()
        d.B, d.Hc, d.W_ = coarse.B, coarse.H, coarse.W
        d.PHc, d.PWc, d.PHs, d.PWs, d.PHo, d.PWo = coarse.PH, coarse.PW, skip.PH, skip.PW, dst.PH, dst.PW
        d.act, d.slope = ACT["leaky"], slope
        self.desc = d
        self.numels = (coarse.t.numel() if coarse.t is not None else 0, skip.t.numel() if skip.t is not None else 0,
                       dst.t.numel() if dst.t is not None else 0)
        self.flops = 2.0 * dst.B * dst.H * dst.W * 32 * (6 * 64 + 9 * 2)
        self.tile = None

    def run(self, A0: torch.Tensor, A1: torch.Tensor, *, out_h: torch.Tensor, stream: Optional[int] = None):
        d = self.desc
        for t, n in zip((A0, A1, out_h), self.numels):
            assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous() and (n == 0 or t.numel() == n)
        d.Xc, d.Xs, d.W, d.bias, d.out_h = A0.data_ptr(), A1.data_ptr(), self.w.data_ptr(), self.bias.data_ptr(), out_h.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().advh_upconv21_tile_f16(C.byref(d), stream), "advh_upconv21_tile_f16")


# ------------------------------------------------------------------------------------------ CPU replay
def replay_on_cpu(plan: GemmPlan, A0: torch.Tensor, A1: Optional[torch.Tensor], out_numel: int,
                  resid: Optional[torch.Tensor] = None, out_init: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Execute a plan's descriptor in numpy exactly as the kernel addresses memory (fp32 math).
    Host-logic test aid: checks ktab / strides / windows without a GPU.  Small shapes only."""
    d = plan.desc
    srcs = [A0.reshape(-1).float().numpy(), None if A1 is None else A1.reshape(-1).float().numpy()]
    W = plan.w.float().numpy()
    if d.wide:                                     # undo the row permutation of the wide packing
        src = packed_row_channel(W.shape[1])
        Wl = np.zeros_like(W)
        Wl[:, src] = W
        W = Wl
    bias = None if plan.bias is None else plan.bias.numpy().reshape(-1)
    out = np.full(out_numel, np.nan, dtype=np.float32) if out_init is None else out_init.reshape(-1).float().numpy().copy()
    kt = plan.ktab_host
    sel, off = (kt >> 31).astype(np.int64), (kt & 0x7FFFFFFF).astype(np.int64)
    for z in range(d.nz):
        for m in range(d.M):
            w_, t_ = m % d.Wg, m // d.Wg
            h_, b_ = t_ % d.Hg, t_ // d.Hg
            ok = d.h0 <= h_ < d.h1 and d.w0 <= w_ < d.w1
            if not ok and not d.halo_zero:
                continue
            zh, zw = (z // d.nz_lo, z % d.nz_lo) if d.nz_lo > 1 else (z, 0)
            orow = b_ * d.o_sB + h_ * d.o_sH + w_ * d.o_sW + d.o_c0 + d.o_sZ * zh + d.o_sZ2 * zw
            cols = np.arange(d.N)
            qn = cols // d.n_div
            hi = (qn // d.n_sub) * d.o_sNhh + (qn % d.n_sub) * d.o_sNhi if d.n_sub > 1 else qn * d.o_sNhi
            o = orow + hi + cols % d.n_div
            if not ok:
                out[o] = 0.0
                continue
            row = np.empty(d.Ktot, dtype=np.float32)
            for c in range(d.Ktot // 8):
                s = sel[c]
                rb = b_ * d.a_sB[s] + h_ * d.a_sH[s] + w_ * d.a_sW[s] + d.a_c0[s] + d.a_sZ[s] * zh + d.a_sZ2[s] * zw
                a = (rb + off[c]) * 8
                row[c * 8:(c + 1) * 8] = srcs[s][a:a + 8]
            v = W[z, :d.N] @ row
            if bias is not None:
                v = v + bias[d.bias_sZ * z: d.bias_sZ * z + d.N]
            if d.act == 1:
                v = torch.nn.functional.gelu(torch.from_numpy(v)).numpy()
            elif d.act == 2:
                v = np.where(v > 0, v, d.slope * v)
            if resid is not None:
                v = v + resid.reshape(-1).float().numpy()[o]
            if d.ph_r > 0:
                to = w_ * d.ph_r + cols // d.n_div - d.ph_pad
                keep = (to >= 0) & (to < d.ph_T)
                o, v = o[keep], v[keep]
            out[o] = v
    return torch.from_numpy(out)


# ------------------------------------------------------------------------------------------ 1-D maps (HiFi-GAN)
@dataclass
class Map1D:
    """Zero-haloed channels-last fp16 sequence ``[B, T + 2*halo, C]`` (C % 8 == 0)."""
    B: int
    T: int
    C: int
    halo: int
    t: Optional[torch.Tensor] = None
    split: bool = False          # fp32-class mode: ``t`` is the plane pair [2, B, P, C]

    @property
    def P(self):
        return self.T + 2 * self.halo

    def alloc(self, device):
        shape = (self.B, self.P, self.C)
        self.t = torch.zeros(((2,) + shape) if self.split else shape, dtype=torch.float16, device=device)
        return self

    def interior(self) -> torch.Tensor:
        t = self.t[0] if self.split else self.t
        return t[:, self.halo:self.halo + self.T]


def plan_conv1d_same(src: Map1D, dst: Map1D, weight: torch.Tensor, bias: Optional[torch.Tensor], *, dilation: int = 1,
                     act: str = "none", slope: float = 0.0, slope2: float = 0.0, device=None) -> GemmPlan:
    """nn.Conv1d with "same" zero padding ((k-1)*d/2 each side) on zero-haloed channels-last maps: the HiFi-GAN
    conv_pre and ResBlock1 convolutions.  Enumerates ``dst``'s padded rows and writes its halo as zeros."""
    Cout, Cin, k = weight.shape
    pad = (k - 1) * dilation // 2
    assert src.C == Cin and Cin % 8 == 0 and src.halo >= pad and (src.B, src.T) == (dst.B, dst.T) and Cout <= dst.C
    cc = Cin // 8
    kt = (np.arange(k)[:, None] * dilation * cc + np.arange(cc)[None, :]).reshape(-1).astype(np.int64)
    w2 = weight.permute(0, 2, 1).reshape(1, Cout, k * Cin).float()
    assert src.split == dst.split
    return GemmPlan(M=dst.B * dst.P, N=Cout, w2=w2, ktab=kt,
                    sources=[Source(src.P * cc, 0, cc, (src.halo - dst.halo - pad) * cc)], Hg=1, Wg=dst.P,
                    window=(0, 1, dst.halo, dst.halo + dst.T), halo_zero=True, out=(dst.P * dst.C, 0, dst.C, 0),
                    bias=bias, act=act, slope=slope, slope2=slope2, device=device, split=dst.split)



# ------------------------------------------------------------------------------------ LDS line-tile conv (narrow C)
class TapsDesc(C.Structure):
    """Mirror of ``advh_taps_desc`` (include/addvisor_hip.h)."""
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p),
                ("out_h", C.c_void_p), ("out_h2", C.c_void_p),
                ("M", C.c_int), ("Hg", C.c_int), ("Wg", C.c_int), ("h0", C.c_int), ("h1", C.c_int),
                ("w0", C.c_int), ("w1", C.c_int), ("ntap", C.c_int), ("toff", C.c_int * 16),
                ("act", C.c_int), ("slope", C.c_float), ("slope2", C.c_float), ("pre_act", C.c_int), ("pre_slope", C.c_float)]


TAPS_MAX_LDS = 160 * 1024


def taps_tile(Cn: int, ntap: int, span: int) -> int:
    """Host copy of ``advh_conv_taps_tile``: positions per workgroup tile (64 * column tiles per wavefront), the
    widest whose weights + two line buffers fit in LDS; 0 = unsupported."""
    for nj in (4, 3, 2):
        if ntap * Cn * Cn * 2 + 2 * (((64 * nj + span) * (Cn // 8) + 63) // 64 * 64) * 16 <= TAPS_MAX_LDS:
            return 64 * nj
    return 0


def taps_lds_bytes(Cn: int, ntap: int, span: int) -> int:
    """Host copy of ``advh_conv_taps_lds_bytes`` (weights + double-buffered line buffer)."""
    tt = taps_tile(Cn, ntap, span)
    return -1 if not tt else ntap * Cn * Cn * 2 + 2 * (((tt + span) * (Cn // 8) + 63) // 64 * 64) * 16


class TapsPlan:
    """One launch of ``advh_conv_taps_f16``: a "same" convolution expressed as taps at constant row offsets of one
    zero-haloed channels-last geometry shared by input, residual and outputs."""

    def __init__(self, *, M: int, Cn: int, w_taps: torch.Tensor, toff: Sequence[int], Hg: int, Wg: int,
                 window: Tuple[int, int, int, int], bias: Optional[torch.Tensor], act: str = "none",
                 slope: float = 0.0, slope2: float = 0.0, device=None, pre_slope: Optional[float] = None, split: bool = False):
        """``split``: the fp32-class form (``advh_conv_taps_split``: every map and the weights are split-format plane pairs, the weights
        stream through LDS tap by tap; 64 channels, no ``pre_slope``)."""
        ntap = len(toff)
        assert w_taps.shape == (ntap, Cn, Cn) and Cn in (32, 64) and 0 < ntap <= 16
        span = max(max(toff), 0) - min(min(toff), 0)
        self.split = split
        if split:
            assert pre_slope is None and taps_split_tile(Cn, ntap, span) > 0
        else:
            assert taps_tile(Cn, ntap, span) > 0
        self.Cn, self.device = Cn, device
        self.w = split_planes(w_taps.double()).contiguous() if split else w_taps.to(torch.float16).contiguous()     # [2, ntap, C, C] | [ntap, C, C]
        self.bias = None if bias is None else bias.to(torch.float32).contiguous()
        if device is not None:
            self.w = self.w.to(device)
            self.bias = None if self.bias is None else self.bias.to(device)
        d = TapsDesc()
        d.M, d.Hg, d.Wg = M, Hg, Wg
        d.h0, d.h1, d.w0, d.w1 = window
        d.ntap = ntap
        for i, t in enumerate(toff):
            d.toff[i] = int(t)
        d.act, d.slope, d.slope2 = ACT[act], slope, slope2
        d.pre_act, d.pre_slope = int(pre_slope is not None), float(pre_slope or 0.0)
        self.desc = d
        valid = (window[1] - window[0]) * (window[3] - window[2]) * (M // (Hg * Wg))
        self.flops = 2.0 * valid * Cn * Cn * ntap

    def run(self, A0: torch.Tensor, A1=None, *, out_h: torch.Tensor, resid: Optional[torch.Tensor] = None,
            out_h2: Optional[torch.Tensor] = None, stream: Optional[int] = None):
        d = self.desc
        planes = 2 if self.split else 1
        for t in (A0, out_h, resid, out_h2):
            assert t is None or (t.dtype == torch.float16 and t.is_cuda and t.numel() == planes * d.M * self.Cn and t.is_contiguous())
        d.X, d.W = A0.data_ptr(), self.w.data_ptr()
        d.bias = self.bias.data_ptr() if self.bias is not None else None
        d.resid = resid.data_ptr() if resid is not None else None
        d.out_h = out_h.data_ptr()
        d.out_h2 = out_h2.data_ptr() if out_h2 is not None else None
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        if self.split:
            n = d.M * self.Cn                                    # plane pairs [2, M, C]: the lo plane one plane behind
            _lib.check(_lib.lib().advh_conv_taps_split(C.byref(d), self.Cn, n, self.w.stride(0), n if resid is not None else 0, n, stream),
                       "advh_conv_taps_split")
        else:
            _lib.check(_lib.lib().advh_conv_taps_f16(C.byref(d), self.Cn, stream), "advh_conv_taps_f16")


class Taps2dDesc(C.Structure):
    """Mirror of ``advh_taps2d_desc`` (include/addvisor_hip.h)."""
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out_h", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W_", C.c_int), ("PH", C.c_int), ("PW", C.c_int),
                ("act", C.c_int), ("slope", C.c_float)]


def taps2d_supported(srcs: Sequence[FMap], dst: FMap, weight: torch.Tensor, stride=(1, 1), padding=(1, 1),
                     dilation=(1, 1)) -> bool:
    """3x3 stride-1 undilated "same" convolution with 32 or 64 channels in and out, one source, source and destination
    maps of one geometry: the layers ``advh_conv_taps2d_f16`` takes."""
    if len(srcs) != 1 or tuple(weight.shape[2:]) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) != (1, 1) \
            or tuple(dilation) != (1, 1):
        return False
    s, Cn = srcs[0], weight.shape[0]
    return (Cn in (32, 64) and weight.shape[1] == Cn and s.C == Cn and dst.C == Cn and s.PH >= 1 and s.PW >= 1
            and (s.B, s.H, s.W, s.PH, s.PW) == (dst.B, dst.H, dst.W, dst.PH, dst.PW))


class Taps2dPlan:
    """One launch of ``advh_conv_taps2d_f16`` (same call shape as ``GemmPlan.run`` for the U-Net step list)."""

    def __init__(self, src: FMap, dst: FMap, weight: torch.Tensor, bias: Optional[torch.Tensor], *, act: str = "leaky",
                 slope: float = 0.2, device=None):
        assert taps2d_supported([src], dst, weight)
        Cn = weight.shape[0]
        self.Cn = Cn
        self.w = weight.permute(2, 3, 0, 1).reshape(9, Cn, Cn).to(torch.float16).contiguous()      # [kh*3+kw][co][ci]
        self.bias = None if bias is None else bias.to(torch.float32).contiguous()
        if device is not None:
            self.w = self.w.to(device)
            self.bias = None if self.bias is None else self.bias.to(device)
        d = Taps2dDesc()
        d.B, d.H, d.W_, d.PH, d.PW = dst.B, dst.H, dst.W, dst.PH, dst.PW
        d.act, d.slope = ACT[act], slope
        self.desc = d
        self.flops = 2.0 * dst.B * dst.H * dst.W * Cn * Cn * 9

    def run(self, A0: torch.Tensor, A1=None, *, out_h: torch.Tensor, stream: Optional[int] = None):
        d = self.desc
        n = d.B * (d.H + 2 * d.PH) * (d.W_ + 2 * d.PW) * self.Cn
        for t in (A0, out_h):
            assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous() and t.numel() == n
        d.X, d.W, d.out_h = A0.data_ptr(), self.w.data_ptr(), out_h.data_ptr()
        d.bias = self.bias.data_ptr() if self.bias is not None else None
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().advh_conv_taps2d_f16(C.byref(d), self.Cn, stream), "advh_conv_taps2d_f16")


class ResblockDesc(C.Structure):
    """Mirror of ``advh_resblock_desc`` (include/addvisor_hip.h)."""
    _fields_ = [("X", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("out_h", C.c_void_p), ("M", C.c_int), ("Wg", C.c_int), ("w0", C.c_int), ("w1", C.c_int), ("k", C.c_int),
                ("dil", C.c_int), ("slope", C.c_float)]


def resblock_pair_lds_bytes(Cn: int, k: int, dil: int) -> int:
    """Host copy of ``advh_resblock_pair_lds_bytes``: both weight tensors + one line buffer (which later holds the
    intermediate tile) if that fits 80 KiB (two workgroups per CU), else two line buffers."""
    rows = max(256 + (k - 1) * dil, 272)
    buf = ((rows * (Cn // 8) + 63) // 64 * 64) * 16
    one = 2 * k * Cn * Cn * 2 + buf
    return one if (one <= 80 * 1024 or one + buf > TAPS_MAX_LDS) else one + buf


def resblock_pair_supported(src: Map1D, dst: Map1D, w1: torch.Tensor, w2: torch.Tensor, dilation: int) -> bool:
    Cn, k = w1.shape[0], w1.shape[2]
    return (Cn in (32, 64) and tuple(w1.shape) == (Cn, Cn, k) and tuple(w2.shape) == (Cn, Cn, k) and k % 2 == 1 and src.C == Cn
            and dst.C == Cn and (src.B, src.T, src.halo) == (dst.B, dst.T, dst.halo) and src.halo >= (k - 1) * dilation // 2
            and resblock_pair_lds_bytes(Cn, k, dilation) <= TAPS_MAX_LDS)


class ResblockPairPlan:
    """One launch of ``advh_resblock_pair_f16``: ``dst = src + conv2(lrelu(conv1(lrelu(src))))`` (HiFi-GAN ResBlock1 step)."""

    def __init__(self, src: Map1D, dst: Map1D, w1, b1, w2, b2, *, dilation: int, slope: float, device=None):
        assert resblock_pair_supported(src, dst, w1, w2, dilation)
        Cn, k = w1.shape[0], w1.shape[2]
        self.Cn = Cn
        pack = lambda w: w.permute(2, 0, 1).to(torch.float16).contiguous()
        self.w1, self.w2 = pack(w1), pack(w2)
        self.b1, self.b2 = b1.to(torch.float32).contiguous(), b2.to(torch.float32).contiguous()
        if device is not None:
            self.w1, self.w2, self.b1, self.b2 = (t.to(device) for t in (self.w1, self.w2, self.b1, self.b2))
        d = ResblockDesc()
        d.M, d.Wg, d.w0, d.w1, d.k, d.dil, d.slope = dst.B * dst.P, dst.P, dst.halo, dst.halo + dst.T, k, dilation, slope
        self.desc = d
        self.flops = 2 * 2.0 * dst.B * dst.T * Cn * Cn * k

    def run(self, A0: torch.Tensor, A1=None, *, out_h: torch.Tensor, stream: Optional[int] = None, **_):
        d = self.desc
        for t in (A0, out_h):
            assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous() and t.numel() == d.M * self.Cn
        assert A0.data_ptr() != out_h.data_ptr()
        d.X, d.out_h = A0.data_ptr(), out_h.data_ptr()
        d.W1, d.W2, d.b1, d.b2 = self.w1.data_ptr(), self.w2.data_ptr(), self.b1.data_ptr(), self.b2.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().advh_resblock_pair_f16(C.byref(d), self.Cn, stream), "advh_resblock_pair_f16")


class ResblockX3Desc(C.Structure):
    """Mirror of ``advh_resblock_x3_desc`` (include/addvisor_hip.h)."""
    _fields_ = [("X", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("out_h", C.c_void_p), ("M", C.c_int), ("Wg", C.c_int), ("w0", C.c_int), ("w1", C.c_int), ("k", C.c_int),
                ("dil", C.c_int), ("slope", C.c_float), ("x_lo", C.c_int64), ("o_lo", C.c_int64), ("w_lo", C.c_int64)]


def resblock_pair_x3_lds_bytes(Cn: int, k: int, dil: int) -> int:
    """Host copy of ``advh_resblock_pair_x3_lds_bytes``: two weight tensors x two planes + one or two split line buffers; -1 = no."""
    if Cn != 32 or k < 1 or k % 2 == 0 or k > 15 or dil < 1:
        return -1
    rows = max(256 + (k - 1) * dil, 272)
    buf = 2 * ((rows * 4 + 63) // 64 * 64) * 16
    one = 4 * k * 32 * 32 * 2 + buf
    lds = one if (one <= 80 * 1024 or one + buf > TAPS_MAX_LDS) else one + buf
    return lds if lds <= TAPS_MAX_LDS else -1


def resblock_pair_x3_supported(src: Map1D, dst: Map1D, w1: torch.Tensor, w2: torch.Tensor, dilation: int) -> bool:
    Cn, k = w1.shape[0], w1.shape[2]
    return (src.split and dst.split and Cn == 32 and tuple(w1.shape) == (Cn, Cn, k) and tuple(w2.shape) == (Cn, Cn, k) and k % 2 == 1
            and src.C == Cn and dst.C == Cn and (src.B, src.T, src.halo) == (dst.B, dst.T, dst.halo)
            and src.halo >= (k - 1) * dilation // 2 and resblock_pair_x3_lds_bytes(Cn, k, dilation) > 0)


class ResblockPairX3Plan:
    """One launch of ``advh_resblock_pair_x3``: ``dst = src + conv2(lrelu(conv1(lrelu(src))))`` on split-format maps (the fp32-class
    form of ``ResblockPairPlan``; HiFi-GAN ResBlock1 step of the 32-channel stage)."""

    def __init__(self, src: Map1D, dst: Map1D, w1, b1, w2, b2, *, dilation: int, slope: float, device=None):
        assert resblock_pair_x3_supported(src, dst, w1, w2, dilation)
        Cn, k = w1.shape[0], w1.shape[2]
        self.Cn = Cn
        pack = lambda w: split_planes(w.permute(2, 0, 1)).contiguous()          # [2][k][C_out][C_in]
        self.w1, self.w2 = pack(w1), pack(w2)
        self.b1, self.b2 = b1.to(torch.float32).contiguous(), b2.to(torch.float32).contiguous()
        if device is not None:
            self.w1, self.w2, self.b1, self.b2 = (t.to(device) for t in (self.w1, self.w2, self.b1, self.b2))
        d = ResblockX3Desc()
        d.M, d.Wg, d.w0, d.w1, d.k, d.dil, d.slope = dst.B * dst.P, dst.P, dst.halo, dst.halo + dst.T, k, dilation, slope
        d.w_lo = k * Cn * Cn
        self.desc = d
        self.flops = 2 * 2.0 * dst.B * dst.T * Cn * Cn * k
        self.tile = None

    def run(self, A0: torch.Tensor, A1=None, *, out_h: torch.Tensor, stream: Optional[int] = None, **_):
        d = self.desc
        for t in (A0, out_h):
            assert t.dtype == torch.float16 and t.is_cuda and t.shape[0] == 2 and t[0].is_contiguous() and t[0].numel() == d.M * self.Cn
        assert A0.data_ptr() != out_h.data_ptr()
        d.X, d.out_h, d.x_lo, d.o_lo = A0.data_ptr(), out_h.data_ptr(), A0.stride(0), out_h.stride(0)
        d.W1, d.W2, d.b1, d.b2 = self.w1.data_ptr(), self.w2.data_ptr(), self.b1.data_ptr(), self.b2.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().advh_resblock_pair_x3(C.byref(d), self.Cn, stream), "advh_resblock_pair_x3")


def taps_split_tile(Cn: int, ntap: int, span: int) -> int:
    """Host copy of ``advh_conv_taps_split_tile``: positions per tile of the fp32-class 64-channel kernel (four weight slots and one line buffer, two planes each, in 160 KiB); 0 = unsupported."""
    if Cn != 64 or not 0 < ntap <= 16:
        return 0
    return 256 if 4 * 2 * 64 * 64 * 2 + 2 * (((256 + span) * 8 + 63) // 64 * 64) * 16 <= 160 * 1024 else 0


def taps_split_supported(src: Map1D, dst: Map1D, weight: torch.Tensor, dilation: int, min_k: int = 7) -> bool:
    """Layers the split-arithmetic line-tile kernel takes AND wins on: 64 channels in and out, k >= ``min_k`` -- 7 for a ResBlock's second
    convolution (residual + two outputs: at k = 3 it is HBM-bound on its 4-byte maps and the implicit GEMM is level), 3 for the first
    (1.07 against 1.37 ms per layer and 256-clip batch at k = 3)."""
    Cout, Cin, k = weight.shape
    return (src.split and dst.split and Cout == Cin == 64 and src.C == 64 and dst.C == 64 and src.halo == dst.halo and min_k <= k <= 16
            and taps_split_tile(64, k, (k - 1) * dilation) > 0)


def taps_supported(src: Map1D, dst: Map1D, weight: torch.Tensor, dilation: int) -> bool:
    Cout, Cin, k = weight.shape
    return (Cout == Cin and Cin in (32, 64) and src.C == Cin and dst.C == Cout and src.halo == dst.halo and k <= 16
            and taps_tile(Cin, k, (k - 1) * dilation) > 0)


def plan_conv1d_taps(src: Map1D, dst: Map1D, weight: torch.Tensor, bias: Optional[torch.Tensor], *, dilation: int = 1,
                     act: str = "none", slope: float = 0.0, slope2: float = 0.0, device=None,
                     pre_slope: Optional[float] = None) -> TapsPlan:
    """Same layer as :func:`plan_conv1d_same`, on the weights-in-LDS kernel (C_in = C_out in {32, 64}).  With
    ``pre_slope`` the input map is the RAW map and LeakyReLU(pre_slope) is applied inside the line buffer."""
    Cout, Cin, k = weight.shape
    pad = (k - 1) * dilation // 2
    split = bool(src.split)
    assert (taps_split_supported(src, dst, weight, dilation, min_k=1) if split else taps_supported(src, dst, weight, dilation))
    assert src.halo >= pad and (src.B, src.T) == (dst.B, dst.T)
    toff = [j * dilation - pad for j in range(k)]
    return TapsPlan(M=dst.B * dst.P, Cn=Cin, w_taps=weight.permute(2, 0, 1).float(), toff=toff, Hg=1, Wg=dst.P,
                    window=(0, 1, dst.halo, dst.halo + dst.T), bias=bias, act=act, slope=slope, slope2=slope2,
                    device=device, pre_slope=pre_slope, split=split)


def replay_taps_on_cpu(plan: TapsPlan, X: torch.Tensor, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Execute a taps descriptor in numpy as the kernel addresses memory (rows clamped to the map).  Test aid."""
    d, Cn = plan.desc, plan.Cn
    x = X.reshape(-1, Cn).float().numpy()
    W = plan.w.float().cpu().numpy()
    out = np.zeros((d.M, Cn), dtype=np.float32)
    rows = np.arange(d.M)
    w_, h_ = rows % d.Wg, (rows // d.Wg) % d.Hg
    ok = (h_ >= d.h0) & (h_ < d.h1) & (w_ >= d.w0) & (w_ < d.w1)
    if d.pre_act:
        x = np.where(x > 0, x, np.float32(np.float16(d.pre_slope)) * x).astype(np.float16).astype(np.float32)
    for t in range(d.ntap):
        src = np.clip(rows + d.toff[t], 0, d.M - 1)
        out += x[src] @ W[t].T
    if plan.bias is not None:
        out += plan.bias.cpu().numpy()[None]
    if d.act == 2:
        out = np.where(out > 0, out, d.slope * out)
    if resid is not None:
        out += resid.reshape(-1, Cn).float().numpy()
    out[~ok] = 0.0
    return torch.from_numpy(out)


def plan_convT1d(src: Map1D, dst: Map1D, weight: torch.Tensor, bias: torch.Tensor, *, stride: int,
                 slope2: float = 0.0, device=None) -> GemmPlan:
    """nn.ConvTranspose1d(k = 2*stride, stride, padding = stride/2) -- the HiFi-GAN upsamplers -- by phase
    decomposition: output ``t = q*r + phase - pad`` sees exactly the two inputs ``x[q-1], x[q]``, so the layer is
    a GEMM over input positions q in [0, T_in] with K = 2*Cin (two adjacent channels-last rows = one contiguous
    slab) and N = r*Cout (phase-major).  Only valid outputs are written; ``dst``'s halo must already be zero."""
    Cin, Cout, k = weight.shape
    r = stride
    pad = (k - r) // 2
    assert k == 2 * r and src.C == Cin and Cin % 8 == 0 and Cout % 4 == 0 and src.halo >= 1
    assert dst.T == src.T * r and dst.C == Cout and dst.halo >= pad and dst.B == src.B
    cc = Cin // 8
    # w2[n = phase*Cout + co][(tap, ci)]: tap 0 = x[q-1] pairs with W[.., phase + r], tap 1 = x[q] with W[.., phase]
    wk = weight.permute(2, 1, 0)                                   # [k, Cout, Cin]
    w2 = torch.cat([wk[r:2 * r], wk[0:r]], dim=2).reshape(1, r * Cout, 2 * Cin).float()
    return GemmPlan(M=src.B * (src.T + 1), N=r * Cout, w2=w2, ktab=np.arange(2 * cc, dtype=np.int64),
                    sources=[Source(src.P * cc, 0, cc, (src.halo - 1) * cc)], Hg=1, Wg=src.T + 1,
                    window=(0, 1, 0, src.T + 1), halo_zero=False,
                    out=(dst.P * dst.C, 0, r * dst.C, (dst.halo - pad) * dst.C), n_div=Cout, o_sNhi=dst.C,
                    bias=bias.float().repeat(r), slope2=slope2, phase=(r, pad, dst.T), device=device, split=dst.split)
