"""The ADDvisor U-Net mask decoder (addvisor.py:12-84) on the HIP kernels.

Feature maps are zero-haloed NHWC fp16 (H = frequency bins, W = frames); every Conv2d /
ConvTranspose2d is one implicit-GEMM launch (``gemm.plan_conv2d`` / ``plan_convT2d``) with BatchNorm
(eval mode) folded into the weights, LeakyReLU(0.2) in the epilogue and the skip
concatenations done by pointer.  The 1-channel stem and the 1x1 mask head are direct kernels.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib, gemm as G

SLOPE = 0.2
BN_EPS = 1e-5


def _fold_bn(sd, conv: str, bn: str, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode BatchNorm folded into the convolution; ``dtype=torch.float64`` for the fp32-class mode (the folded
    weight is then split into two fp16 planes from its fp64 value)."""
    w, b = sd[conv + ".weight"].to(dtype), sd[conv + ".bias"].to(dtype)
    s = sd[bn + ".weight"].to(dtype) / torch.sqrt(sd[bn + ".running_var"].to(dtype) + BN_EPS)
    return w * s.view(-1, 1, 1, 1), (b - sd[bn + ".running_mean"].to(dtype)) * s + sd[bn + ".bias"].to(dtype)


def reference_flops(B: int, H: int, W: int) -> float:
    """2 x MACs of addvisor.py:31-60 on a ``B x 1 x H x W`` input (convolutions, transposed convolutions, 1x1 head)."""
    def conv(h, w, cin, cout, taps):
        return 2.0 * B * h * w * cin * cout * taps
    f = conv(H // 2, W, 1, 32, 15) + conv(H // 2, W, 32, 32, 9)
    f += conv(H // 4, W, 32, 64, 15) + conv(H // 4, W, 64, 64, 9)
    f += conv(H // 8, W // 2, 64, 128, 9) + conv(H // 8, W // 2, 128, 128, 9)
    f += conv(H // 16, W // 4, 128, 256, 9) + conv(H // 16, W // 4, 256, 256, 9)
    f += conv(H // 16, W // 4, 256, 512, 9) + conv(H // 16, W // 4, 512, 512, 9)
    f += conv(H // 16, W // 4, 512, 256, 4) + conv(H // 8, W // 2, 384, 256, 9) + conv(H // 8, W // 2, 256, 256, 9)
    f += conv(H // 8, W // 2, 256, 128, 4) + conv(H // 4, W, 192, 128, 9) + conv(H // 4, W, 128, 128, 9)
    f += conv(H // 4, W, 128, 64, 2) + conv(H // 2, W, 96, 64, 9) + conv(H // 2, W, 64, 64, 9)
    f += conv(H // 2, W, 64, 32, 2) + conv(H, W, 33, 32, 9) + conv(H, W, 32, 32, 9)
    return f + conv(H, W, 32, 1, 1)


class HipUNet:
    """``forward(mag [B, F>=H, T>=W] fp32) -> mask [B, H, W] fp32`` (H % 16 == 0, W % 4 == 0)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, line_tile: Optional[bool] = None, fuse_up: Optional[bool] = None,
                 precision: Optional[str] = None):
        """``line_tile``: run the 3x3 32- / 64-channel same-geometry layers on the weights-in-LDS kernel
        (``advh_conv_taps2d_f16``) instead of the implicit GEMM.  ``fuse_up``: fold every ConvTranspose2d into the
        convolution that follows it (``gemm.plan_upconv2d``) so the up-sampled maps are never written.
        ``precision``: "f16" | "f32" (default ``ADDVISOR_PRECISION``, i.e. f32): "f32" is the fp32-class mode -- every map a
        split-format plane pair, every convolution three MFMAs per product (``advh_gemm_desc.split``), weights folded in
        fp64 -- whose ``mask > 0.5`` index set reproduces the reference's fp32 CPU result (addvisor.py:57-60; asserted
        against tests/golden/unet.npz).  The line-tile kernels are fp16-only, so that mode runs the implicit GEMM."""
        _lib.init()
        from .embedder import default_precision
        self.precision = precision or default_precision()
        if self.precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.split = self.precision == "f32"
        self.wdtype = torch.float64 if self.split else torch.float32
        if self.split:
            line_tile = False
        if line_tile is None:                                  # both default on; the environment switches are for A/B measurements
            line_tile = os.environ.get("ADDVISOR_UNET_LINE_TILE", "1") != "0"
        if fuse_up is None:
            fuse_up = os.environ.get("ADDVISOR_UNET_FUSE_UP", "1") != "0"
        self.dev, self.line_tile, self.fuse_up = device, line_tile, fuse_up
        self.sd = {k.replace("module.", ""): v.detach() for k, v in sd.items()}     # LMAC_metrics.py:23-25
        w, b = _fold_bn(self.sd, "e1.block.0", "e1.block.1", self.wdtype)
        self.stem_w = w.reshape(32, 15).float().contiguous().to(device)
        self.stem_b = b.float().contiguous().to(device)
        self.head_w = self.sd["mask_head.0.weight"].float().reshape(32).contiguous().to(device)
        self.head_b = float(self.sd["mask_head.0.bias"].float().reshape(-1)[0])
        self._ws: Dict[Tuple[int, int, int], dict] = {}

    def _workspace(self, B: int, H: int, W: int) -> dict:
        key = (B, H, W)
        if key in self._ws:
            return self._ws[key]
        if H % 16 or W % 4:
            raise ValueError("U-Net input needs H % 16 == 0 and W % 4 == 0 (SURVEY.md D2)")
        dev, sd = self.dev, self.sd
        F = lambda h, w, c, ph, pw: G.FMap(B, h, w, c, ph, pw, split=self.split).alloc(dev)
        if self.fuse_up:
            return self._workspace_fused(B, H, W)
        m = dict(
            x1a=F(H // 2, W, 32, 2, 1), x1=F(H // 2, W, 32, 2, 1),
            x2a=F(H // 4, W, 64, 1, 1), x2=F(H // 4, W, 64, 1, 1),
            x3a=F(H // 8, W // 2, 128, 1, 1), x3=F(H // 8, W // 2, 128, 1, 1),
            x4a=F(H // 16, W // 4, 256, 1, 1), x4=F(H // 16, W // 4, 256, 2, 2),
            b1=F(H // 16, W // 4, 512, 4, 4), b2=F(H // 16, W // 4, 512, 0, 0),
            u4=F(H // 8, W // 2, 256, 1, 1), y4a=F(H // 8, W // 2, 256, 1, 1), y4=F(H // 8, W // 2, 256, 0, 0),
            u3=F(H // 4, W, 128, 1, 1), y3a=F(H // 4, W, 128, 1, 1), y3=F(H // 4, W, 128, 0, 0),
            u2=F(H // 2, W, 64, 1, 1), y2a=F(H // 2, W, 64, 1, 1), y2=F(H // 2, W, 64, 1, 1),
            u1=F(H, W, 40, 1, 1), y1a=F(H, W, 32, 1, 1), y1=F(H, W, 32, 1, 1),
        )
        steps = []

        def conv(srcs, dst, conv_name, bn_name, **kw):
            w, b = _fold_bn(sd, conv_name, bn_name, self.wdtype)
            cin = sum(m[s].C for s in srcs)
            if cin != w.shape[1]:                              # d1: 33 real channels live in a 40-wide map
                w = torch.cat([w, w.new_zeros(w.shape[0], cin - w.shape[1], *w.shape[2:])], 1)
            if self.line_tile and G.taps2d_supported([m[s] for s in srcs], m[dst], w, **kw):
                plan = G.Taps2dPlan(m[srcs[0]], m[dst], w, b, slope=SLOPE, device=dev)
            else:
                plan = G.plan_conv2d([m[s] for s in srcs], m[dst], w, b, slope=SLOPE, device=dev, **kw)
            steps.append((plan, srcs, dst))

        def block(srcs, mid, dst, name, **first):              # ConvBlock, addvisor.py:12-25
            conv(srcs, mid, f"{name}.block.0", f"{name}.block.1", **first)
            conv([mid], dst, f"{name}.block.3", f"{name}.block.4")

        def up(src, dst, name, stride):
            plan = G.plan_convT2d(m[src], m[dst], sd[name + ".weight"].to(self.wdtype), sd[name + ".bias"].float(),
                                  stride=stride, device=dev)
            steps.append((plan, [src], dst))

        conv(["x1a"], "x1", "e1.block.3", "e1.block.4")        # e1.block.0 is the direct stem kernel
        block(["x1"], "x2a", "x2", "e2", stride=(2, 1), padding=(2, 1))
        block(["x2"], "x3a", "x3", "e3", stride=(2, 2))
        block(["x3"], "x4a", "x4", "e4", stride=(2, 2))
        conv(["x4"], "b1", "bottleneck.0", "bottleneck.1", padding=(2, 2), dilation=(2, 2))
        conv(["b1"], "b2", "bottleneck.3", "bottleneck.4", padding=(4, 4), dilation=(4, 4))
        up("b2", "u4", "up4", (2, 2))
        block(["u4", "x3"], "y4a", "y4", "d4")
        up("y4", "u3", "up3", (2, 2))
        block(["u3", "x2"], "y3a", "y3", "d3")
        up("y3", "u2", "up2", (2, 1))
        block(["u2", "x1"], "y2a", "y2", "d2")
        up("y2", "u1", "up1", (2, 1))
        block(["u1"], "y1a", "y1", "d1")
        ws = dict(maps=m, steps=steps, mask=torch.empty(B, H, W, dtype=torch.float32, device=dev),
                  logits=torch.empty(B, H, W, dtype=torch.float32, device=dev))
        ws["flops"] = reference_flops(B, H, W)
        self._ws[key] = ws
        return ws

    def _workspace_fused(self, B: int, H: int, W: int) -> dict:
        """Same network with up4+d4, up3+d3, up2+d2, up1+d1 as fused launches: the maps u4..u1 do not exist.  The coarse
        maps b2 / y4 / y3 carry one extra 8-channel chunk whose first channel is the in-image indicator (and unused
        channels up to the next multiple of 64, so every pixel starts on a 128-byte line); for d1 the
        indicator rides in the 8-channel spectrogram map ``xin`` = (x, 1, 0, ...) that ``advh_unet_pack_x`` fills."""
        dev, sd = self.dev, self.sd
        F = lambda h, w, c, ph, pw: G.FMap(B, h, w, c, ph, pw, split=self.split).alloc(dev)
        m = dict(
            x1a=F(H // 2, W, 32, 2, 1), x1=F(H // 2, W, 32, 2, 1),
            x2a=F(H // 4, W, 64, 1, 1), x2=F(H // 4, W, 64, 1, 1),
            x3a=F(H // 8, W // 2, 128, 1, 1), x3=F(H // 8, W // 2, 128, 1, 1),
            x4a=F(H // 16, W // 4, 256, 1, 1), x4=F(H // 16, W // 4, 256, 2, 2),
            b1=F(H // 16, W // 4, 512, 4, 4), b2=G.add_indicator(F(H // 16, W // 4, 576, 1, 1), 512),
            y4a=F(H // 8, W // 2, 256, 1, 1), y4=G.add_indicator(F(H // 8, W // 2, 320, 1, 1), 256),
            y3a=F(H // 4, W, 128, 1, 1), y3=G.add_indicator(F(H // 4, W, 192, 1, 1), 128),
            y2a=F(H // 2, W, 64, 1, 1), y2=F(H // 2, W, 64, 1, 1),
            xin=G.add_indicator(F(H, W, 8, 1, 1), 1), y1a=F(H, W, 32, 1, 1), y1=F(H, W, 32, 1, 1),
        )
        steps = []

        def conv(srcs, dst, conv_name, bn_name, **kw):
            w, b = _fold_bn(sd, conv_name, bn_name, self.wdtype)
            if self.line_tile and G.taps2d_supported([m[s] for s in srcs], m[dst], w, **kw):
                plan = G.Taps2dPlan(m[srcs[0]], m[dst], w, b, slope=SLOPE, device=dev)
            elif self.line_tile and G.conv_s21_supported([m[s] for s in srcs], m[dst], w, **kw):
                plan = G.ConvS21TilePlan(m[srcs[0]], m[dst], w, b, slope=SLOPE, device=dev)
            else:
                plan = G.plan_conv2d([m[s] for s in srcs], m[dst], w, b, slope=SLOPE, device=dev, **kw)
            steps.append((plan, srcs, dst))

        def block(srcs, mid, dst, name, **first):
            conv(srcs, mid, f"{name}.block.0", f"{name}.block.1", **first)
            conv([mid], dst, f"{name}.block.3", f"{name}.block.4")

        def up_block(coarse, skip, mid, dst, up_name, name, stride, coarse_C, skip_C, indicator):
            wc, bc = _fold_bn(sd, f"{name}.block.0", f"{name}.block.1", self.wdtype)
            wt, bt = sd[up_name + ".weight"].to(self.wdtype), sd[up_name + ".bias"].to(self.wdtype)
            if self.line_tile and G.upconv_tile_supported(m[coarse], m[skip], m[mid], wt, wc, stride, indicator):
                plan = G.UpconvTilePlan(m[coarse], m[skip], m[mid], wt, bt, wc, bc, slope=SLOPE, device=dev)
            else:
                plan = G.plan_upconv2d(m[coarse], m[skip], m[mid], wt, bt, wc, bc, stride=stride, coarse_C=coarse_C,
                                       skip_C=skip_C, indicator=indicator, slope=SLOPE, device=dev)
            steps.append((plan, [coarse, skip], mid))
            conv([mid], dst, f"{name}.block.3", f"{name}.block.4")

        conv(["x1a"], "x1", "e1.block.3", "e1.block.4")
        block(["x1"], "x2a", "x2", "e2", stride=(2, 1), padding=(2, 1))
        block(["x2"], "x3a", "x3", "e3", stride=(2, 2))
        block(["x3"], "x4a", "x4", "e4", stride=(2, 2))
        conv(["x4"], "b1", "bottleneck.0", "bottleneck.1", padding=(2, 2), dilation=(2, 2))
        conv(["b1"], "b2", "bottleneck.3", "bottleneck.4", padding=(4, 4), dilation=(4, 4))
        up_block("b2", "x3", "y4a", "y4", "up4", "d4", (2, 2), 512, 128, ("coarse", 512))
        up_block("y4", "x2", "y3a", "y3", "up3", "d3", (2, 2), 256, 64, ("coarse", 256))
        up_block("y3", "x1", "y2a", "y2", "up2", "d2", (2, 1), 128, 32, ("coarse", 128))
        up_block("y2", "xin", "y1a", "y1", "up1", "d1", (2, 1), 64, 1, ("skip", 1))
        ws = dict(maps=m, steps=steps, mask=torch.empty(B, H, W, dtype=torch.float32, device=dev),
                  logits=torch.empty(B, H, W, dtype=torch.float32, device=dev))
        ws["flops"] = reference_flops(B, H, W)                # the reference formulation's count, not the fused launches'
        self._ws[(B, H, W)] = ws
        return ws

    def flops(self, B: int, H: int, W: int) -> float:
        return self._workspace(B, H, W)["flops"]

    def forward(self, mag: torch.Tensor, H: int = 512, W: Optional[int] = None, want_logits: bool = False):
        """``mag``: fp32 CUDA ``[B, Fq, Tq]`` (t fastest); the ``H x W`` crop is read in place."""
        if mag.dim() != 3 or mag.dtype != torch.float32 or not mag.is_cuda:
            raise ValueError("mag must be a CUDA fp32 tensor [B, F, T]")
        mag = mag.contiguous()
        B, Fq, Tq = mag.shape
        W = (Tq // 4) * 4 if W is None else W
        if H > Fq or W > Tq:
            raise ValueError("crop exceeds the spectrogram")
        ws = self._workspace(B, H, W)
        m, lib = ws["maps"], _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        x1a = m["x1a"]
        xcat, xc0 = (m["xin"], 0) if self.fuse_up else (m["u1"], 32)
        if self.split:
            _lib.check(lib.advh_unet_stem_split(mag.data_ptr(), Fq, Tq, B, H, W, self.stem_w.data_ptr(), self.stem_b.data_ptr(),
                                                x1a.t.data_ptr(), x1a.t.stride(0), x1a.PH, x1a.PW, SLOPE, st), "advh_unet_stem_split")
            _lib.check(lib.advh_unet_pack_x_split(mag.data_ptr(), Fq, Tq, B, H, W, xcat.t.data_ptr(), xcat.t.stride(0), xcat.C, xc0,
                                                  xcat.PH, xcat.PW, st), "advh_unet_pack_x_split")
        else:
            _lib.check(lib.advh_unet_stem(mag.data_ptr(), Fq, Tq, B, H, W, self.stem_w.data_ptr(), self.stem_b.data_ptr(),
                                          x1a.t.data_ptr(), x1a.PH, x1a.PW, SLOPE, st), "advh_unet_stem")
            _lib.check(lib.advh_unet_pack_x(mag.data_ptr(), Fq, Tq, B, H, W, xcat.t.data_ptr(), xcat.C, xc0, xcat.PH, xcat.PW, st),
                       "advh_unet_pack_x")
        for plan, srcs, dst in ws["steps"]:
            a0 = m[srcs[0]].t
            a1 = m[srcs[1]].t if len(srcs) > 1 else None
            plan.run(a0, a1, out_h=m[dst].t)
        y1 = m["y1"]
        if self.split:
            _lib.check(lib.advh_unet_head_split(y1.t.data_ptr(), y1.t.stride(0), B, H, W, y1.PH, y1.PW, self.head_w.data_ptr(),
                                                self.head_b, ws["mask"].data_ptr(), ws["logits"].data_ptr(), st), "advh_unet_head_split")
        else:
            _lib.check(lib.advh_unet_head(y1.t.data_ptr(), B, H, W, y1.PH, y1.PW, self.head_w.data_ptr(), self.head_b,
                                          ws["mask"].data_ptr(), ws["logits"].data_ptr(), st), "advh_unet_head")
        mask = ws["mask"].clone()
        return (mask, ws["logits"].clone()) if want_logits else mask
