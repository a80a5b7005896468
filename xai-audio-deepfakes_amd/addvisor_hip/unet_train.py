"""Training step of the ADDvisor U-Net mask decoder on the HIP kernels.

Reference: ``mask = model(magnitude)`` in ``train()`` mode and ``loss.backward()`` through it
(train_addvisor.py:364-378 over addvisor.py:12-84): Conv2d -> BatchNorm2d (batch statistics) -> LeakyReLU(0.2)
blocks, ConvTranspose2d upsamplers, skip concatenations, 1x1 sigmoid head.

Forward   every Conv2d is the implicit GEMM of ``gemm.plan_conv2d`` writing the raw pre-BatchNorm map ``z`` (fp16,
          zero halo); ``advh_bn_stats`` + ``advh_bn_apply`` turn it into the activation map the next layer reads.
          Weights are re-packed on the device from the live fp32 parameters at every call (``GemmPlan.load_weights``).
Backward  per layer, in reverse:
          * BatchNorm + LeakyReLU backward (``advh_bn_bwd_sums`` / ``advh_bn_bwd_apply``) -> ``dz`` (fp16, scaled),
            written straight into the geometry its consumers read (zero-upsampled grid for the strided layers);
          * dgrad = the same implicit GEMM with transposed, flipped weights (a strided layer's dgrad is a stride-1
            convolution over the zero-upsampled ``dz``); skip gradients accumulate in place through ``resid``;
          * wgrad: the reduction runs over positions, so both operands are transposed to position-major once
            (``advh_transpose_gather``: the horizontal taps become extra rows, the vertical taps become 8-aligned
            K offsets because the common grid's width is a multiple of 8) and the product is a split-K launch of the
            same GEMM kernel (``w_ld`` / grid-z batches) with fp32 partial outputs summed afterwards.  3x3 stride-1 layers skip all
            of that: ``advh_conv_wgrad2d_f16`` / ``advh_conv_wgrad2d_split`` read both position-major maps through the transposing
            LDS load and keep the nine taps' accumulators in registers (fp16: square 32 / 64-channel layers; fp32-class: every layer
            with 32-multiple channel counts, one launch per (source slice, output slice) pair).
          ConvTranspose2d (kernel = stride): dgrad is a strided convolution, wgrad the same transposed GEMM with the
          sub-pixel taps gathered by the transpose.
Gradients between kernels are fp16 scaled by a power of two chosen from the incoming mask gradient; sums are fp32/fp64.

Precision (``precision=``, default ``ADDVISOR_PRECISION`` = f32): the reference trains in fp32 (train_addvisor.py:363-378).
"f32" = the fp32-class mode: every map above is a split-format plane pair ``[2, B, Hp, Wp, C]`` (hi + lo * 2^-11, ~22 bits),
the forward convolutions, dgrads and split-K wgrads are the three-MFMA launches (``desc.split``), the BatchNorm / head / stem
kernels read and write plane pairs, the operand transposes (strided / dilated / up-sampling layers only) run once per plane -- so
LeakyReLU takes the branch the fp32 reference takes and parameter gradients agree with fp32 autograd to ~1e-5 instead of the fp16
mode's cosine 0.98.  "f16" = fp16 maps and gradients (half the bytes, a third of the MFMAs).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, gemm as G
from .embedder import default_precision

SLOPE = 0.2
WGRAD_TILES = True   # 3x3 square 32/64-channel layers: wgrad on the LDS-tile kernel (ds_read_b64_tr_b16; fp16 and split-arithmetic forms) instead of transposes + split-K GEMM
G32 = False       # keep the activation gradients a BatchNorm backward consumes in fp32 (measured: no accuracy difference, +30 % BN traffic)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class MapGeom(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("PH", C.c_int), ("PW", C.c_int)]


class Wgrad2dDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("DZ", C.c_void_p), ("partial", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W_", C.c_int),
                ("PHx", C.c_int), ("PWx", C.c_int), ("PHz", C.c_int), ("PWz", C.c_int)]


class TransposeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Hg", C.c_int), ("Wg", C.c_int), ("GH", C.c_int), ("GW", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("Hs", C.c_int), ("Ws", C.c_int), ("PHs", C.c_int), ("PWs", C.c_int), ("Cs", C.c_int), ("c0", C.c_int), ("nC", C.c_int),
                ("sy", C.c_int), ("sx", C.c_int), ("ntap", C.c_int), ("oy", C.c_int * 16), ("ox", C.c_int * 16),
                ("ld", C.c_int64), ("col0", C.c_int64), ("rpt", C.c_int), ("r0", C.c_int)]


def _geom(f: G.FMap) -> MapGeom:
    return MapGeom(f.B, f.H, f.W, f.C, f.PH, f.PW)


def _st():
    return torch.cuda.current_stream().cuda_stream


# (conv name, bn name, source maps, destination map, kernel, stride, padding, dilation); "up" rows: (name, src, dst, stride)
_CONVS = [
    ("conv", "e1.block.0", "e1.block.1", ["mag"], "x1a", (5, 3), (2, 1), (2, 1), (1, 1)),
    ("conv", "e1.block.3", "e1.block.4", ["x1a"], "x1", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "e2.block.0", "e2.block.1", ["x1"], "x2a", (5, 3), (2, 1), (2, 1), (1, 1)),
    ("conv", "e2.block.3", "e2.block.4", ["x2a"], "x2", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "e3.block.0", "e3.block.1", ["x2"], "x3a", (3, 3), (2, 2), (1, 1), (1, 1)),
    ("conv", "e3.block.3", "e3.block.4", ["x3a"], "x3", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "e4.block.0", "e4.block.1", ["x3"], "x4a", (3, 3), (2, 2), (1, 1), (1, 1)),
    ("conv", "e4.block.3", "e4.block.4", ["x4a"], "x4", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "bottleneck.0", "bottleneck.1", ["x4"], "b1", (3, 3), (1, 1), (2, 2), (2, 2)),
    ("conv", "bottleneck.3", "bottleneck.4", ["b1"], "b2", (3, 3), (1, 1), (4, 4), (4, 4)),
    ("up", "up4", "b2", "u4", (2, 2)),
    ("conv", "d4.block.0", "d4.block.1", ["u4", "x3"], "y4a", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "d4.block.3", "d4.block.4", ["y4a"], "y4", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("up", "up3", "y4", "u3", (2, 2)),
    ("conv", "d3.block.0", "d3.block.1", ["u3", "x2"], "y3a", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "d3.block.3", "d3.block.4", ["y3a"], "y3", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("up", "up2", "y3", "u2", (2, 1)),
    ("conv", "d2.block.0", "d2.block.1", ["u2", "x1"], "y2a", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "d2.block.3", "d2.block.4", ["y2a"], "y2", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("up", "up1", "y2", "u1", (2, 1)),
    ("conv", "d1.block.0", "d1.block.1", ["u1"], "y1a", (3, 3), (1, 1), (1, 1), (1, 1)),
    ("conv", "d1.block.3", "d1.block.4", ["y1a"], "y1", (3, 3), (1, 1), (1, 1), (1, 1)),
]


def _conv_w2(w: torch.Tensor, splits: List[int]) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, KH, KW] -> the [1, Cout, sum_s KH*KW*C_s] operand of ``plan_conv2d`` (per source:
    taps outer, channels inner; sources concatenated), zero-padding Cin up to ``sum(splits)``."""
    Cout, Cin = w.shape[:2]
    tot = sum(splits)
    if tot != Cin:
        w = torch.cat([w, w.new_zeros(Cout, tot - Cin, *w.shape[2:])], 1)
    parts, lo = [], 0
    for c in splits:
        parts.append(w[:, lo:lo + c].permute(0, 2, 3, 1).reshape(Cout, -1))
        lo += c
    return torch.cat(parts, 1)[None]


class _SplitKGemm:
    """Weight-gradient product ``out[z][m][n] = sum_{k in slice z} A[m][k + off] * W[n][k]`` on ``advh_gemm_f16``:
    A and W are position-major fp16 matrices on the device (``advh_transpose_gather`` outputs)."""

    def __init__(self, Mrows: int, N: int, Kc: int, nz: int, a_ld: int, a_col: int, w_ld: int, device, groups: int = 1,
                 group_off: int = 0, split: bool = False):
        """``groups`` > 1: the A rows come in ``groups`` blocks of ``Mrows`` (the vertical taps of a convolution); block
        ``i`` reads the same matrix rows shifted by ``i * group_off`` columns (the tap's K offset), so all taps are one launch."""
        assert Kc % G.BK == 0 and a_ld % 8 == 0 and a_col % 8 == 0 and a_col >= 0 and w_ld % 8 == 0 and group_off % 8 == 0
        self.tile, _ = G.pick_tile(N, Mrows * groups)
        self.ktab = torch.arange(Kc // 8, dtype=torch.int32, device=device)
        d = G.GemmDesc()
        d.M, d.N, d.Ktot, d.w_rows = Mrows * groups, N, Kc, N
        d.Hg, d.Wg = groups, Mrows
        d.h0, d.h1, d.w0, d.w1 = 0, groups, 0, Mrows
        d.halo_zero = 0
        d.a_sB[0], d.a_sH[0], d.a_sW[0], d.a_c0[0], d.a_sZ[0] = 0, group_off // 8, a_ld // 8, a_col // 8, Kc // 8
        d.w_sZ, d.bias_sZ, d.w_ld = Kc, 0, w_ld
        d.o_sB, d.o_sH, d.o_sW, d.o_c0, d.o_sNhi, d.o_sZ = 0, Mrows * N, N, 0, 0, Mrows * groups * N
        d.n_div, d.nz, d.act, d.slope = G.round_up(N, 4), nz, 0, 0.0
        d.ktab_identity, d.wide = 1, 0
        d.split = int(split)
        self.split = split
        self.desc, self.flops = d, 2.0 * Mrows * groups * N * Kc * nz

    def run(self, A: torch.Tensor, W: torch.Tensor, out_f: torch.Tensor):
        d = self.desc
        if self.split:                                     # plane pairs [2, rows, ld]: lo plane = stride(0) elements behind
            assert A.shape[0] == 2 and W.shape[0] == 2 and A.stride(0) % 8 == 0
            d.a_lo[0], d.w_lo, d.o_lo = A.stride(0) // 8, W.stride(0), 0
        d.A0, d.A1, d.W, d.ktab = A.data_ptr(), None, W.data_ptr(), self.ktab.data_ptr()
        d.bias = d.resid = d.out_h = d.out_h2 = d.out_pre = d.dact_src = None
        d.out_f = out_f.data_ptr()
        _lib.check(_lib.lib().advh_gemm_f16(C.byref(d), self.tile, _st()), "advh_gemm_f16 (wgrad)")


def _split_k(Mrows: int, N: int, Mg: int) -> Tuple[int, int]:
    """(nz, Kc): enough K slices for ~1500 workgroups, each slice a multiple of 64 positions."""
    _, BN = G.pick_tile(N, Mrows)
    BM = 128 if BN == 128 else 256
    tiles = -(-Mrows // BM) * -(-N // BN)
    nz = max(1, min(-(-1536 // tiles), Mg // 4096 if Mg >= 8192 else 1))
    Kc = G.round_up(-(-Mg // nz), G.BK)
    return nz, Kc


class HipUNetTrain:
    """``forward(mag) -> mask`` with batch-statistics BatchNorm, then ``backward(dmask) -> {param name: grad}``.

    ``params``: live fp32 device tensors keyed like the module's ``state_dict`` (weights, biases, BatchNorm affine and
    running buffers); they are read at every ``forward`` and the running statistics are updated in place."""

    def __init__(self, params: Dict[str, torch.Tensor], device, precision: Optional[str] = None):
        _lib.init()
        self.dev, self.p = device, params
        self.precision = precision or default_precision()
        if self.precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.split = self.precision == "f32"
        self.nparts = _lib.lib().advh_bn_partial_count()
        self._ws: Dict[Tuple[int, int, int], dict] = {}
        self._last = None
        self.generation = 0                                    # bumped by every forward: backward() belongs to the latest one

    # ------------------------------------------------------------------------------------------ workspace
    def _workspace(self, B: int, H: int, W: int) -> dict:
        key = (B, H, W)
        if key in self._ws:
            return self._ws[key]
        if H % 16 or W % 4:
            raise ValueError("U-Net input needs H % 16 == 0 and W % 4 == 0 (SURVEY.md D2)")
        dev, p, sp = self.dev, self.p, self.split
        pl = (2,) if sp else ()
        F = lambda h, w, c, ph, pw: G.FMap(B, h, w, c, ph, pw, split=sp).alloc(dev)
        geo = dict(                                            # activation maps: the geometry of addvisor_hip/unet.py
            x1a=(H // 2, W, 32, 1, 1), x1=(H // 2, W, 32, 2, 1), x2a=(H // 4, W, 64, 1, 1), x2=(H // 4, W, 64, 1, 1),
            x3a=(H // 8, W // 2, 128, 1, 1), x3=(H // 8, W // 2, 128, 1, 1), x4a=(H // 16, W // 4, 256, 1, 1),
            x4=(H // 16, W // 4, 256, 2, 2), b1=(H // 16, W // 4, 512, 4, 4), b2=(H // 16, W // 4, 512, 0, 0),
            u4=(H // 8, W // 2, 256, 1, 1), y4a=(H // 8, W // 2, 256, 1, 1), y4=(H // 8, W // 2, 256, 0, 0),
            u3=(H // 4, W, 128, 1, 1), y3a=(H // 4, W, 128, 1, 1), y3=(H // 4, W, 128, 0, 0),
            u2=(H // 2, W, 64, 1, 1), y2a=(H // 2, W, 64, 1, 1), y2=(H // 2, W, 64, 0, 0),
            u1=(H, W, 40, 1, 1), y1a=(H, W, 32, 1, 1), y1=(H, W, 32, 0, 0))
        m = {k: F(*v) for k, v in geo.items()}                 # activations
        z, g = {}, {}
        for k, (h, w, c, ph, pw) in geo.items():                # gradient w.r.t. the activation map: fp32 where a BatchNorm
            f = G.FMap(B, h, w, 32 if k == "u1" else c, ph, pw, split=sp)  # backward consumes it (its mean is subtracted there),
            f.t = torch.zeros(pl + (B, f.Hp, f.Wp, f.C), dtype=torch.float16 if (sp or k.startswith("u") or not G32) else torch.float32, device=dev)
            g[k] = f                                             # fp16 where it is only a GEMM operand (the upsampled maps)
        layers = []
        for row in _CONVS:
            if row[0] == "up":
                _, name, src, dst, stride = row
                layers.append(self._plan_up(name, m[src], m[dst], g[src], g[dst], stride, B))
                continue
            _, cname, bname, srcs, dst, k, stride, pad, dil = row
            h, w, c, ph, pw = geo[dst]
            z[dst] = F(h, w, c, ph, pw)
            layers.append(self._plan_conv(cname, bname, srcs, dst, k, stride, pad, dil, m, z, g, B, H, W))
        ws = dict(maps=m, z=z, g=g, layers=layers,
                  mask=torch.empty(B, H, W, dtype=torch.float32, device=dev), logits=torch.empty(B, H, W, dtype=torch.float32, device=dev),
                  dlogit=torch.empty(B, H, W, dtype=torch.float32, device=dev),
                  partial=torch.empty(self.nparts * 2 * 512, dtype=torch.float32, device=dev),
                  sums=torch.empty(2 * 512, dtype=torch.float32, device=dev))
        self._ws[key] = ws
        return ws

    def _plan_conv(self, cname, bname, srcs, dst, k, stride, pad, dil, m, z, g, B, H, W) -> dict:
        dev, p, sp = self.dev, self.p, self.split
        pl = (2,) if sp else ()
        (KH, KW), (sh, sw), (ph, pw), (dh, dw) = k, stride, pad, dil
        L = dict(kind="conv", cname=cname, bname=bname, srcs=srcs, dst=dst, k=k, stride=stride, pad=pad, dil=dil)
        Cout = p[cname + ".weight"].shape[0]
        stem = srcs == ["mag"]
        if not stem:
            splits = [m[s].C for s in srcs]
            L["splits"] = splits
            w2 = _conv_w2(p[cname + ".weight"].detach(), splits)
            L["fwd"] = G.plan_conv2d([m[s] for s in srcs], z[dst], torch.zeros(Cout, sum(splits), KH, KW), torch.zeros(Cout),
                                     stride=stride, padding=pad, dilation=dil, act="none", device=dev)
            assert w2.shape[2] == L["fwd"].K
        # ---- backward geometry.  dz lives on the grid the dgrad / wgrad read: the output grid for stride 1, the
        # INPUT-size grid with dz scattered at (sh*h, sw*w) for a strided layer; halo = the dgrad convolution's padding
        Hd, Wd = (m[dst].H, m[dst].W) if (sh, sw) == (1, 1) else (m[dst].H * sh, m[dst].W * sw)
        if stem:
            L["dz"] = G.FMap(B, m[dst].H, m[dst].W, Cout, 0, 0, split=sp).alloc(dev)         # dense: only the stem wgrad reads it
            L["dz_strides"] = (L["dz"].Hp * L["dz"].Wp * Cout, L["dz"].Wp * Cout, Cout, 0)
            L["stem_part"] = torch.empty(self.nparts * 480, dtype=torch.float32, device=dev)
            return L
        pph, ppw = dh * (KH - 1) - ph, dw * (KW - 1) - pw
        dzm = G.FMap(B, Hd, Wd, Cout, pph, ppw, split=sp).alloc(dev)
        L["dz"] = dzm
        L["dz_strides"] = (dzm.Hp * dzm.Wp * Cout, sh * dzm.Wp * Cout, sw * Cout, (pph * dzm.Wp + ppw) * Cout)
        # ---- dgrad plans (one per source; the 40-channel d1 concat map only needs its 32 up-sampled channels)
        L["dgrad"] = []
        lo = 0
        for s in srcs:
            c = g[s].C
            plan = G.plan_conv2d([dzm], g[s], torch.zeros(c, Cout, KH, KW), None, stride=(1, 1), padding=(pph, ppw), dilation=dil,
                                 act="none", device=dev)
            L["dgrad"].append((plan, s, lo, c))
            lo += m[s].C
        Cin = sum(m[s].C for s in srcs)
        L["Cin"] = Cin
        # ---- wgrad, 3x3 stride-1 layers: LDS-tile kernel with transposing operand reads, no operand copies.  fp16: square 32 / 64-channel
        # layers; fp32-class: every layer whose sources and output have a multiple of 32 channels, one launch per (source slice, output slice)
        # pair of 32 / 64 channels (csrc/conv_wgrad.hip)
        f0 = m[srcs[0]]
        plain33 = WGRAD_TILES and k == (3, 3) and stride == (1, 1) and dil == (1, 1)
        if (plain33 and not sp and len(srcs) == 1 and Cin == Cout and Cout in (32, 64) and f0.PH >= 1 and f0.PW >= 1):
            nparts = _lib.lib().advh_conv_wgrad2d_parts(Cout, B, Hd, Wd)
            L["wg2d"] = Wgrad2dDesc(B=B, H=Hd, W_=Wd, PHx=f0.PH, PWx=f0.PW, PHz=pph, PWz=ppw)
            L["wg2d_part"] = torch.empty(nparts * 9 * Cout * Cout, dtype=torch.float32, device=dev)
            return L
        # the d1 concat map "u1" is 32 up-sampled channels + the magnitude (channel 32) + 7 zero channels: its 32-channel part is a slice like
        # any other, the magnitude channel has its own kernel (advh_unet_skip_wgrad, reading the fp32 input directly)
        skip_src = srcs == ["u1"] and m["u1"].C == 40 and Cout == 32
        if (plain33 and sp and Cout % 32 == 0 and pph >= 1 and ppw >= 1
                and all((m[s_].C % 32 == 0 or skip_src) and m[s_].PH >= 1 and m[s_].PW >= 1 for s_ in srcs)):
            cuts = lambda n: [(o, 64) for o in range(0, n - n % 64, 64)] + ([(n - 32, 32)] if n % 64 else [])
            pairs, base, nmax = [], 0, 0
            for s_ in srcs:
                f = m[s_]
                d2 = Wgrad2dDesc(B=B, H=Hd, W_=Wd, PHx=f.PH, PWx=f.PW, PHz=pph, PWz=ppw)
                for cx0, CI in cuts(32 if skip_src else f.C):
                    for cz0, CO in cuts(Cout):
                        pairs.append((s_, d2, CI, CO, cx0, cz0, base))
                        nmax = max(nmax, _lib.lib().advh_conv_wgrad2d_split_parts(CI, CO, B, Hd, Wd) * 9 * CI * CO)
                base += f.C
            L["wg2d_pairs"] = pairs
            L["wg2d_part"] = torch.empty(nmax, dtype=torch.float32, device=dev)
            if skip_src:
                L["wg2d_skip"] = torch.empty(self.nparts * 288, dtype=torch.float32, device=dev)
            return L
        # ---- wgrad: position-major operands on the common grid (dz's interior grid + vertical tap halo, width % 8 == 0)
        Hg, Wg, GH = Hd + (KH - 1) * dh, G.round_up(Wd, 8), ph
        Mg = B * Hg * Wg
        Mrows = KW * Cin
        nz, Kc = _split_k(KH * Mrows, Cout, Mg)
        g_lo, g_hi = ph * Wg, ((KH - 1) * dh - ph) * Wg + 64
        a_ld = g_lo + nz * Kc + g_hi
        L["XT"] = torch.zeros(pl + (Mrows, a_ld), dtype=torch.float16, device=dev)
        L["dzT"] = torch.zeros(pl + (Cout, nz * Kc), dtype=torch.float16, device=dev)
        L["wpart"] = torch.empty(nz, KH, Mrows, Cout, dtype=torch.float32, device=dev)
        L["wg"] = _SplitKGemm(Mrows, Cout, Kc, nz, a_ld, g_lo - ph * Wg, nz * Kc, dev, groups=KH, group_off=dh * Wg, split=sp)
        tds, r0 = [], 0
        for s in srcs:
            f = m[s]
            td = TransposeDesc(B=B, Hg=Hg, Wg=Wg, GH=GH, GW=0, H=Hd, W=Wd, Hs=f.H, Ws=f.W, PHs=f.PH, PWs=f.PW, Cs=f.C, c0=0, nC=f.C,
                               sy=1, sx=1, ntap=KW, ld=a_ld, col0=g_lo, rpt=Cin, r0=r0)
            for j in range(KW):
                td.oy[j], td.ox[j] = 0, j * dw - pw
            tds.append((s, td))
            r0 += f.C
        L["x_tr"] = tds
        td = TransposeDesc(B=B, Hg=Hg, Wg=Wg, GH=GH, GW=0, H=Hd, W=Wd, Hs=Hd, Ws=Wd, PHs=pph, PWs=ppw, Cs=Cout, c0=0, nC=Cout,
                           sy=1, sx=1, ntap=1, ld=nz * Kc, col0=0, rpt=Cout, r0=0)
        L["dz_tr"] = td
        return L

    def _plan_up(self, name, src: G.FMap, dst: G.FMap, gsrc: G.FMap, gdst: G.FMap, stride, B) -> dict:
        dev, p, sp = self.dev, self.p, self.split
        pl = (2,) if sp else ()
        sh, sw = stride
        Cin, Cout = p[name + ".weight"].shape[:2]
        L = dict(kind="up", name=name, stride=stride, src=src, dst=dst, gsrc=gsrc, gdst=gdst, Cin=Cin, Cout=Cout)
        L["fwd"] = G.plan_convT2d(src, dst, torch.zeros(Cin, Cout, sh, sw), torch.zeros(Cout), stride=stride, device=dev)
        # dgrad: a stride = kernel convolution over the gradient of the upsampled map
        L["dgrad"] = G.plan_conv2d([gdst], gsrc, torch.zeros(Cin, Cout, sh, sw), None, stride=stride, padding=(0, 0), act="none", device=dev)
        Hg, Wg = src.H, G.round_up(src.W, 8)
        Mg = B * Hg * Wg
        N = sh * sw * Cout
        nz, Kc = _split_k(Cin, N, Mg)
        L["XT"] = torch.zeros(pl + (Cin, nz * Kc + 64), dtype=torch.float16, device=dev)
        L["GT"] = torch.zeros(pl + (N, nz * Kc), dtype=torch.float16, device=dev)
        L["wpart"] = torch.empty(nz, Cin, N, dtype=torch.float32, device=dev)
        L["wg"] = _SplitKGemm(Cin, N, Kc, nz, nz * Kc + 64, 0, nz * Kc, dev, split=sp)
        L["x_tr"] = TransposeDesc(B=B, Hg=Hg, Wg=Wg, GH=0, GW=0, H=src.H, W=src.W, Hs=src.H, Ws=src.W, PHs=src.PH, PWs=src.PW,
                                  Cs=src.C, c0=0, nC=Cin, sy=1, sx=1, ntap=1, ld=nz * Kc + 64, col0=0, rpt=Cin, r0=0)
        td = TransposeDesc(B=B, Hg=Hg, Wg=Wg, GH=0, GW=0, H=src.H, W=src.W, Hs=gdst.H, Ws=gdst.W, PHs=gdst.PH, PWs=gdst.PW,
                           Cs=gdst.C, c0=0, nC=Cout, sy=sh, sx=sw, ntap=sh * sw, ld=nz * Kc, col0=0, rpt=Cout, r0=0)
        for i in range(sh):
            for j in range(sw):
                td.oy[i * sw + j], td.ox[i * sw + j] = i, j
        L["g_tr"] = td
        return L

    # ------------------------------------------------------------------------------------------ forward
    def _bn_forward(self, L, zmap: G.FMap, amap: G.FMap, ws):
        lib, p, st = _lib.lib(), self.p, _st()
        Cn = zmap.C
        gm = _geom(zmap)
        self._bn_stats(zmap, ws)
        n = float(zmap.B * zmap.H * zmap.W)
        gamma, beta = p[L["bname"] + ".weight"].detach(), p[L["bname"] + ".bias"].detach()
        rm, rv, nb = (p.get(L["bname"] + sfx) for sfx in (".running_mean", ".running_var", ".num_batches_tracked"))
        if "coef" not in L:
            L["coef"] = torch.empty(4 * Cn, dtype=torch.float32, device=self.dev)
            L["coef_b"] = torch.empty(3 * Cn, dtype=torch.float32, device=self.dev)
        L["n"] = n
        nbp = nb.data_ptr() if nb is not None and nb.dtype == torch.int64 else None
        _lib.check(lib.advh_bn_coef(ws["sums"].data_ptr(), gamma.data_ptr(), beta.data_ptr(), Cn, n, BN_EPS, BN_MOMENTUM,
                                    None if rm is None else rm.data_ptr(), None if rv is None else rv.data_ptr(), nbp,
                                    L["coef"].data_ptr(), st), "advh_bn_coef")
        if self.split:
            _lib.check(lib.advh_bn_apply_split(zmap.t.data_ptr(), zmap.t.stride(0), C.byref(gm), L["coef"].data_ptr(), SLOPE, amap.t.data_ptr(),
                                               amap.t.stride(0), st), "advh_bn_apply_split")
        else:
            _lib.check(lib.advh_bn_apply(zmap.t.data_ptr(), C.byref(gm), L["coef"].data_ptr(), SLOPE, amap.t.data_ptr(), st), "advh_bn_apply")

    def _bn_stats(self, fmap: G.FMap, ws):
        """per-channel (sum, sum of squares) of a map's interior -> ws["sums"]"""
        lib, st, gm = _lib.lib(), _st(), _geom(fmap)
        if self.split:
            _lib.check(lib.advh_bn_stats_split(fmap.t.data_ptr(), fmap.t.stride(0), C.byref(gm), ws["partial"].data_ptr(),
                                               ws["sums"].data_ptr(), st), "advh_bn_stats_split")
        else:
            _lib.check(lib.advh_bn_stats(fmap.t.data_ptr(), C.byref(gm), ws["partial"].data_ptr(), ws["sums"].data_ptr(), st), "advh_bn_stats")

    def forward(self, mag: torch.Tensor, H: int = 512, W: Optional[int] = None) -> torch.Tensor:
        if mag.dim() != 3 or mag.dtype != torch.float32 or not mag.is_cuda:
            raise ValueError("mag must be a CUDA fp32 tensor [B, F, T]")
        mag = mag.contiguous()
        B, Fq, Tq = mag.shape
        W = (Tq // 4) * 4 if W is None else W
        if H > Fq or W > Tq:
            raise ValueError("crop exceeds the spectrogram")
        ws = self._workspace(B, H, W)
        m, z, lib, p, st, sp = ws["maps"], ws["z"], _lib.lib(), self.p, _st(), self.split
        u1 = m["u1"]
        if sp:
            _lib.check(lib.advh_unet_pack_x_split(mag.data_ptr(), Fq, Tq, B, H, W, u1.t.data_ptr(), u1.t.stride(0), u1.C, 32, u1.PH, u1.PW, st),
                       "advh_unet_pack_x_split")
        else:
            _lib.check(lib.advh_unet_pack_x(mag.data_ptr(), Fq, Tq, B, H, W, u1.t.data_ptr(), u1.C, 32, u1.PH, u1.PW, st), "advh_unet_pack_x")
        for L in ws["layers"]:
            if L["kind"] == "up":
                w = p[L["name"] + ".weight"].detach()
                sh, sw = L["stride"]
                L["fwd"].load_weights(w.permute(2, 3, 1, 0).reshape(1, sh * sw * L["Cout"], L["Cin"]), p[L["name"] + ".bias"].detach().repeat(sh * sw))
                L["fwd"].run(L["src"].t, out_h=L["dst"].t)
                continue
            dst = L["dst"]
            if L["srcs"] == ["mag"]:                             # 1-channel stem: direct kernel, raw weights, identity activation
                sw_ = p[L["cname"] + ".weight"].detach().reshape(32, 15).contiguous()
                sb_ = p[L["cname"] + ".bias"].detach().contiguous()
                L["keep"] = (sw_, sb_)
                if sp:
                    _lib.check(lib.advh_unet_stem_split(mag.data_ptr(), Fq, Tq, B, H, W, sw_.data_ptr(), sb_.data_ptr(), z[dst].t.data_ptr(),
                                                        z[dst].t.stride(0), z[dst].PH, z[dst].PW, 1.0, st), "advh_unet_stem_split")
                else:
                    _lib.check(lib.advh_unet_stem(mag.data_ptr(), Fq, Tq, B, H, W, sw_.data_ptr(), sb_.data_ptr(), z[dst].t.data_ptr(),
                                                  z[dst].PH, z[dst].PW, 1.0, st), "advh_unet_stem")
            else:
                L["fwd"].load_weights(_conv_w2(p[L["cname"] + ".weight"].detach(), L["splits"]), p[L["cname"] + ".bias"].detach())
                srcs = L["srcs"]
                L["fwd"].run(m[srcs[0]].t, m[srcs[1]].t if len(srcs) > 1 else None, out_h=z[dst].t)
            self._bn_forward(L, z[dst], m[dst], ws)
        y1 = m["y1"]
        hw = p["mask_head.0.weight"].detach().reshape(32).contiguous()
        hb = float(p["mask_head.0.bias"].detach().reshape(-1)[0])
        if sp:
            _lib.check(lib.advh_unet_head_split(y1.t.data_ptr(), y1.t.stride(0), B, H, W, y1.PH, y1.PW, hw.data_ptr(), hb, ws["mask"].data_ptr(),
                                                ws["logits"].data_ptr(), st), "advh_unet_head_split")
        else:
            _lib.check(lib.advh_unet_head(y1.t.data_ptr(), B, H, W, y1.PH, y1.PW, hw.data_ptr(), hb, ws["mask"].data_ptr(),
                                          ws["logits"].data_ptr(), st), "advh_unet_head")
        self._last = (mag, B, Fq, Tq, H, W, hw)
        self.generation += 1
        return ws["mask"].clone()

    # ------------------------------------------------------------------------------------------ backward
    def _tr(self, src_t: torch.Tensor, dst: torch.Tensor, td: TransposeDesc):
        if self.split:                                     # a transpose moves elements: once per plane
            for pl in range(2):
                _lib.check(_lib.lib().advh_transpose_gather(src_t[pl].data_ptr(), dst[pl].data_ptr(), C.byref(td), _st()), "advh_transpose_gather")
            return
        _lib.check(_lib.lib().advh_transpose_gather(src_t.data_ptr(), dst.data_ptr(), C.byref(td), _st()), "advh_transpose_gather")

    def backward(self, dmask: torch.Tensor) -> Dict[str, torch.Tensor]:
        """``dmask [B, H, W]`` = dL/d mask of the last ``forward`` -> fp32 gradients of every trainable parameter."""
        mag, B, Fq, Tq, H, W, hw = self._last
        ws = self._workspace(B, H, W)
        m, z, g, lib, p, st, sp = ws["maps"], ws["z"], ws["g"], _lib.lib(), self.p, _st(), self.split
        dmask = dmask.to(self.dev, torch.float32).contiguous()
        assert dmask.shape == (B, H, W)
        peak = float((dmask.abs().max() * 0.25 * hw.abs().max()).item())
        S = 2.0 ** max(-24, min(24, math.floor(math.log2(64.0 / peak)))) if peak > 0 and math.isfinite(peak) else 1.0
        grads: Dict[str, torch.Tensor] = {}
        y1, gy1 = m["y1"], g["y1"]
        dw33 = torch.empty(64, dtype=torch.float32, device=self.dev)
        if sp:
            _lib.check(lib.advh_unet_head_bwd_split(dmask.data_ptr(), ws["mask"].data_ptr(), hw.data_ptr(), S, B * H * W, ws["dlogit"].data_ptr(),
                                                    gy1.t.data_ptr(), gy1.t.stride(0), st), "advh_unet_head_bwd_split")
            _lib.check(lib.advh_unet_head_wgrad_split(ws["dlogit"].data_ptr(), y1.t.data_ptr(), y1.t.stride(0), B * H * W,
                                                      ws["partial"].data_ptr(), dw33.data_ptr(), st), "advh_unet_head_wgrad_split")
        else:
            _lib.check(lib.advh_unet_head_bwd(dmask.data_ptr(), ws["mask"].data_ptr(), hw.data_ptr(), S, B * H * W, ws["dlogit"].data_ptr(),
                                              gy1.t.data_ptr(), int(gy1.t.dtype == torch.float32), st), "advh_unet_head_bwd")
            _lib.check(lib.advh_unet_head_wgrad(ws["dlogit"].data_ptr(), y1.t.data_ptr(), B * H * W, ws["partial"].data_ptr(),
                                                dw33.data_ptr(), st), "advh_unet_head_wgrad")
        grads["mask_head.0.weight"] = dw33[:32].clone().view(1, 32, 1, 1)
        grads["mask_head.0.bias"] = dw33[32:33].clone()
        fresh = set()                                            # skip maps whose gradient has been written once already
        for L in reversed(ws["layers"]):
            if L["kind"] == "up":
                self._up_backward(L, ws, grads, S)
                continue
            dst = L["dst"]
            Cn, n = z[dst].C, L["n"]
            gm = _geom(z[dst])
            coef = L["coef"]
            g32 = int(g[dst].t.dtype == torch.float32)
            if sp:
                _lib.check(lib.advh_bn_bwd_sums_split(z[dst].t.data_ptr(), z[dst].t.stride(0), g[dst].t.data_ptr(), g[dst].t.stride(0), C.byref(gm),
                                                      coef.data_ptr(), SLOPE, ws["partial"].data_ptr(), ws["sums"].data_ptr(), st), "advh_bn_bwd_sums_split")
            else:
                _lib.check(lib.advh_bn_bwd_sums(z[dst].t.data_ptr(), g[dst].t.data_ptr(), g32, C.byref(gm), coef.data_ptr(), SLOPE,
                                                ws["partial"].data_ptr(), ws["sums"].data_ptr(), st), "advh_bn_bwd_sums")
            dgam, dbet = (torch.empty(Cn, dtype=torch.float32, device=self.dev) for _ in range(2))
            coef_b = L["coef_b"]
            _lib.check(lib.advh_bn_bwd_coef(ws["sums"].data_ptr(), coef.data_ptr(), Cn, n, 1.0 / S, coef_b.data_ptr(), dgam.data_ptr(),
                                            dbet.data_ptr(), st), "advh_bn_bwd_coef")
            grads[L["bname"] + ".weight"], grads[L["bname"] + ".bias"] = dgam, dbet
            dzm = L["dz"]
            sB, sH, sW, c0 = L["dz_strides"]
            if sp:
                _lib.check(lib.advh_bn_bwd_apply_split(z[dst].t.data_ptr(), z[dst].t.stride(0), g[dst].t.data_ptr(), g[dst].t.stride(0), C.byref(gm),
                                                       coef.data_ptr(), coef_b.data_ptr(), SLOPE, dzm.t.data_ptr(), dzm.t.stride(0), sB, sH, sW, c0, st),
                           "advh_bn_bwd_apply_split")
            else:
                _lib.check(lib.advh_bn_bwd_apply(z[dst].t.data_ptr(), g[dst].t.data_ptr(), g32, C.byref(gm), coef.data_ptr(), coef_b.data_ptr(),
                                                 SLOPE, dzm.t.data_ptr(), sB, sH, sW, c0, st), "advh_bn_bwd_apply")
            grads[L["cname"] + ".bias"] = torch.zeros_like(p[L["cname"] + ".bias"])    # exactly zero before a batch-stat BatchNorm
            w = p[L["cname"] + ".weight"].detach()
            if L["srcs"] == ["mag"]:
                dw = torch.empty(32, 15, dtype=torch.float32, device=self.dev)
                if sp:
                    _lib.check(lib.advh_unet_stem_wgrad_split(dzm.t.data_ptr(), dzm.t.stride(0), Fq, Tq, B, H, W, mag.data_ptr(), dzm.PH, dzm.PW,
                                                              L["stem_part"].data_ptr(), dw.data_ptr(), st), "advh_unet_stem_wgrad_split")
                else:
                    _lib.check(lib.advh_unet_stem_wgrad(dzm.t.data_ptr(), Fq, Tq, B, H, W, mag.data_ptr(), dzm.PH, dzm.PW,
                                                        L["stem_part"].data_ptr(), dw.data_ptr(), st), "advh_unet_stem_wgrad")
                grads[L["cname"] + ".weight"] = (dw / S).view(32, 1, 5, 3)
                continue
            KH, KW = L["k"]
            # ---- wgrad
            if "wg2d_pairs" in L:
                Cn, Cin = z[dst].C, L["Cin"]
                dwf = torch.empty(Cn, Cin, 3, 3, dtype=torch.float32, device=self.dev)
                for sname, d2, CI, CO, cx0, cz0, base in L["wg2d_pairs"]:
                    xm = m[sname]
                    d2.X, d2.DZ, d2.partial = xm.t.data_ptr(), dzm.t.data_ptr(), L["wg2d_part"].data_ptr()
                    dw9 = torch.empty(9, CO, CI, dtype=torch.float32, device=self.dev)
                    _lib.check(lib.advh_conv_wgrad2d_split(C.byref(d2), CI, CO, xm.C, cx0, Cn, cz0, xm.t.stride(0), dzm.t.stride(0),
                                                            dw9.data_ptr(), st), "advh_conv_wgrad2d_split")
                    dwf[cz0:cz0 + CO, base + cx0:base + cx0 + CI] = dw9.view(3, 3, CO, CI).permute(2, 3, 0, 1)
                if "wg2d_skip" in L:                                   # the magnitude channel of the d1 concat map
                    dws = torch.empty(32, 9, dtype=torch.float32, device=self.dev)
                    _lib.check(lib.advh_unet_skip_wgrad_split(dzm.t.data_ptr(), dzm.t.stride(0), Fq, Tq, B, H, W, mag.data_ptr(), dzm.PH, dzm.PW,
                                                               L["wg2d_skip"].data_ptr(), dws.data_ptr(), st), "advh_unet_skip_wgrad_split")
                    dwf[:, 33:] = 0.0
                    dwf[:, 32] = dws.view(32, 3, 3)
                grads[L["cname"] + ".weight"] = (dwf / S)[:, :w.shape[1]].contiguous()
            elif "wg2d" in L:
                d2 = L["wg2d"]
                d2.X, d2.DZ, d2.partial = m[L["srcs"][0]].t.data_ptr(), dzm.t.data_ptr(), L["wg2d_part"].data_ptr()
                Cn = z[dst].C
                dw9 = torch.empty(9, Cn, Cn, dtype=torch.float32, device=self.dev)
                _lib.check(lib.advh_conv_wgrad2d_f16(C.byref(d2), Cn, dw9.data_ptr(), st), "advh_conv_wgrad2d_f16")
                grads[L["cname"] + ".weight"] = (dw9.view(3, 3, Cn, Cn).permute(2, 3, 0, 1) / S).contiguous()
            else:
                self._wgrad_gemm(L, m, z, dzm, dst, w, grads, S)
            self._dgrad(L, w, g, fresh)
        return grads

    def _wgrad_gemm(self, L, m, z, dzm, dst, w, grads, S):
        KH, KW = L["k"]
        for sname, td in L["x_tr"]:
            self._tr(m[sname].t, L["XT"], td)
        self._tr(dzm.t, L["dzT"], L["dz_tr"])
        L["wg"].run(L["XT"], L["dzT"], L["wpart"])
        Cin, Cout = L["Cin"], z[dst].C
        dw = L["wpart"].sum(0).view(KH, KW, Cin, Cout).permute(3, 2, 0, 1) / S              # [Cout, Cin, KH, KW]
        grads[L["cname"] + ".weight"] = dw[:, :w.shape[1]].contiguous()

    def _dgrad(self, L, w, g, fresh):
        wt = w.permute(1, 0, 2, 3).flip(2, 3)                                               # [Cin, Cout, KH, KW]
        for plan, sname, lo, c in L["dgrad"]:
            plan.load_weights(_conv_w2(wt[lo:lo + c], [wt.shape[1]]))
            tgt = g[sname]
            out = dict(out_h=tgt.t) if tgt.t.dtype == torch.float16 else dict(out_f=tgt.t)
            if sname in fresh:
                plan.run(L["dz"].t, resid=tgt.t, **out)
            else:
                plan.run(L["dz"].t, **out)
                fresh.add(sname)

    def _up_backward(self, L, ws, grads, S):
        lib, p, st = _lib.lib(), self.p, _st()
        name, (sh, sw), Cin, Cout = L["name"], L["stride"], L["Cin"], L["Cout"]
        gd, gs = L["gdst"], L["gsrc"]
        self._bn_stats(gd, ws)
        grads[name + ".bias"] = ws["sums"][:Cout].clone() / S
        self._tr(L["src"].t, L["XT"], L["x_tr"])
        self._tr(gd.t, L["GT"], L["g_tr"])
        L["wg"].run(L["XT"], L["GT"], L["wpart"])
        grads[name + ".weight"] = (L["wpart"].sum(0).view(Cin, sh, sw, Cout).permute(0, 3, 1, 2) / S).contiguous()
        w = p[name + ".weight"].detach()                                                        # [Cin, Cout, sh, sw] = conv weight [Cout'=Cin][Cin'=Cout]
        L["dgrad"].load_weights(_conv_w2(w, [Cout]))
        L["dgrad"].run(gd.t, **(dict(out_f=gs.t) if gs.t.dtype == torch.float32 else dict(out_h=gs.t)))
