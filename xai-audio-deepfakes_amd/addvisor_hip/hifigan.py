"""HiFi-GAN V1 generator on the HIP kernels (reference handle: hifigan.py:106-110, 180).

``decode_batch(mel [B, 80, T]) -> wav [B, 1, T*256]``.  Every Conv1d / ConvTranspose1d is an
implicit-GEMM launch on zero-haloed channels-last fp16 maps (``gemm.plan_conv1d_same`` /
``plan_convT1d``); LeakyReLU is applied by the producer (each GEMM writes the raw map for the residual
path and, through ``out_h2``, the pre-activated copy the next conv reads), the MRF average and the
1-channel conv_post + tanh are small direct kernels.  Weight-norm is assumed folded (inference form).

Fidelity options.  The reference reaches the generator through SpeechBrain's wrapper, whose source is not available
offline; two of its choices change the numbers and are therefore options here, mirrored by ``oracle/hifigan_ref.py``:
``padding_mode`` of the "same" Conv1d layers ("zeros" = the published Kong et al. model, the default; "reflect" = the
default of SpeechBrain's ``Conv1d``) and ``inference_padding`` (mel frames replicated on both sides before the
generator, output length ``(T + 2 p) * 256``; 0 = off, the default; SpeechBrain / Coqui generators ship with 5).
"reflect" fills the map halos with the mirrored interior after every producing launch (``advh_halo_fill_f16``); the
fused line-tile kernels keep their intermediate in LDS with zero padding, so that mode runs the implicit GEMM only.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib, gemm as G
from .embedder import default_precision
from .synthetic import HifiganConfig

HALO = 32          # >= the largest "same" padding: (11 - 1) * 5 / 2 = 25


class HipHifigan:
    def __init__(self, cfg: HifiganConfig, sd: Dict[str, torch.Tensor], device, line_tile: bool = True, fuse: bool = True,
                 padding_mode: str = "zeros", inference_padding: int = 0, precision: Optional[str] = None):
        """``line_tile``: run the 32- / 64-channel ResBlock convolutions on the weights-in-LDS kernel
        (``advh_conv_taps_f16``) instead of the implicit GEMM; ``fuse``: whole ResBlock steps in one kernel where both
        weight tensors fit in LDS (``advh_resblock_pair_f16``).  ``padding_mode`` / ``inference_padding``: see the module
        docstring.  ``precision``: None = ``ADDVISOR_PRECISION`` (default "f32": the fp32-class mode of the explanation path
        -- split-format maps, three MFMAs per product; the reference runs the vocoder in fp32, hifigan.py:180) or "f16"
        (fp16 operands, LDS line-tile kernels; stated tolerance 2e-2 on waveforms)."""
        _lib.init()
        precision = precision or default_precision()
        self._ctor = dict(line_tile=line_tile, fuse=fuse, padding_mode=padding_mode, inference_padding=inference_padding)
        if precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.precision, self.split = precision, precision == "f32"
        # fp32-class mode: the 32-channel ResBlock steps have a fused split-format kernel (csrc/resblock_pair_x3.hip), the k >= 7 convolutions
        # of the 64-channel stage a split-format line tile with streamed weights (csrc/conv_taps.hip); the other layers run the x3 implicit GEMM
        self.fuse_x3 = self.split and fuse and padding_mode == "zeros"
        if self.split:
            line_tile = fuse = False
        if padding_mode not in ("zeros", "reflect"):
            raise ValueError("padding_mode must be 'zeros' or 'reflect'")
        if inference_padding < 0:
            raise ValueError("inference_padding must be >= 0")
        self.padding_mode, self.inference_padding = padding_mode, int(inference_padding)
        if padding_mode == "reflect":
            line_tile = fuse = False                       # the fused kernels zero-pad their LDS intermediate
        self.cfg, self.dev, self.line_tile, self.fuse = cfg, device, line_tile, fuse
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        ch = cfg.upsample_initial_channel
        for _ in cfg.upsample_rates:
            ch //= 2
        if ch % 8:
            raise ValueError("every HiFi-GAN stage needs a channel count that is a multiple of 8")
        if cfg.in_channels % 8:
            raise ValueError("mel channels must be a multiple of 8")
        self.post_w = self.sd["conv_post.weight"][0].t().contiguous().to(device)       # [k][C]
        self.post_b = float(self.sd["conv_post.bias"][0])
        self._ws: Dict[Tuple[int, int], dict] = {}

    def with_precision(self, precision: str) -> "HipHifigan":
        """This generator at ``precision`` (itself if it already is): ``ExplainPipeline`` runs its vocoder at the path's
        precision, so that the explanation has ONE arithmetic class."""
        if precision == self.precision:
            return self
        return type(self)(self.cfg, self.sd, self.dev, precision=precision, **self._ctor)

    def _workspace(self, B: int, T: int) -> dict:
        key = (B, T)
        if key in self._ws:
            return self._ws[key]
        cfg, sd, dev = self.cfg, self.sd, self.dev
        M = lambda t, c: G.Map1D(B, t, c, HALO, split=self.split).alloc(dev)

        def conv(src, dst, w, b, **kw):
            if self.line_tile and G.taps_supported(src, dst, w, kw.get("dilation", 1)):
                return G.plan_conv1d_taps(src, dst, w, b, device=dev, **kw)
            if (self.fuse_x3 and kw.get("pre_slope") is None
                    and G.taps_split_supported(src, dst, w, kw.get("dilation", 1), min_k=3 if kw.get("act") == "leaky" else 7)):
                return G.plan_conv1d_taps(src, dst, w, b, device=dev, **kw)       # fp32-class line tile with streamed weights (64 channels; first convolutions k >= 3, second ones k >= 7)
            if kw.pop("pre_slope", None) is not None:
                raise RuntimeError("a layer of a line-buffer-activated stage does not fit the line-tile kernel")
            return G.plan_conv1d_same(src, dst, w, b, device=dev, **kw)

        ch = cfg.upsample_initial_channel
        reflect = self.padding_mode == "reflect"
        mel = M(T, cfg.in_channels)
        cur = M(T, ch)                       # lrelu(conv_pre(mel))
        steps = []
        if reflect:
            steps.append(("halo", 1, mel, None, None, None))
        steps.append(("gemm", G.plan_conv1d_same(mel, cur, sd["conv_pre.weight"], sd["conv_pre.bias"], act="leaky",
                                                 slope=cfg.leaky_slope, device=dev), mel, None, cur, None))
        t = T
        nk, nd = len(cfg.resblock_kernel_sizes), len(cfg.resblock_dilations)
        nstage = len(cfg.upsample_rates)
        for i, r in enumerate(cfg.upsample_rates):
            co, t2 = ch // 2, t * r
            # stages whose ResBlock convolutions run on the line-tile kernel apply LeakyReLU to the raw map inside the
            # line buffer: no pre-activated copies are stored or read there
            in_lds = self.line_tile and co in (32, 64) and all(
                G.taps_tile(co, k, (k - 1) * max(cfg.resblock_dilations)) > 0 for k in cfg.resblock_kernel_sizes)
            # fp32-class mode: a stage whose EVERY ResBlock step has the fused split-format kernel needs no pre-activated copies either
            x3_stage = self.fuse_x3 and all(G.resblock_pair_x3_lds_bytes(co, k, dd) > 0 for k in cfg.resblock_kernel_sizes
                                            for dd in cfg.resblock_dilations) and HALO >= (max(cfg.resblock_kernel_sizes) - 1) * max(cfg.resblock_dilations) // 2
            in_lds = in_lds or x3_stage
            x = M(t2, co)
            lx = None if in_lds else M(t2, co)
            steps.append(("gemm", G.plan_convT1d(cur, x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=r,
                                                 slope2=cfg.leaky_slope, device=dev), cur, None, x, lx))
            if reflect:                                    # the ResBlock convolutions read x / lrelu(x) reflect-padded
                steps.append(("halo", 1, x, None, None, None))
                steps.append(("halo", 1, lx, None, None, None))
            tmp, pa, pb = M(t2, co), M(t2, co), M(t2, co)
            la, lb = (None, None) if in_lds else (M(t2, co), M(t2, co))
            outs = [M(t2, co) for _ in range(nk)]
            for j in range(nk):
                p = f"resblocks.{i * nk + j}."
                cx, clx = x, lx
                for d in range(nd):
                    last = d == nd - 1
                    ox = outs[j] if last else (pa if d % 2 == 0 else pb)
                    w1, w2 = sd[p + f"convs1.{d}.weight"], sd[p + f"convs2.{d}.weight"]
                    if x3_stage:
                        assert G.resblock_pair_x3_supported(cx, ox, w1, w2, cfg.resblock_dilations[d])
                        steps.append(("gemm", G.ResblockPairX3Plan(cx, ox, w1, sd[p + f"convs1.{d}.bias"], w2, sd[p + f"convs2.{d}.bias"],
                                                                  dilation=cfg.resblock_dilations[d], slope=cfg.leaky_slope, device=dev),
                                      cx, None, ox, None))
                        cx, clx = ox, None
                        continue
                    if in_lds and self.fuse and G.resblock_pair_supported(cx, ox, w1, w2, cfg.resblock_dilations[d]):
                        # both convolutions of the step in one kernel, the intermediate map stays in LDS
                        steps.append(("gemm", G.ResblockPairPlan(cx, ox, w1, sd[p + f"convs1.{d}.bias"], w2, sd[p + f"convs2.{d}.bias"],
                                                                dilation=cfg.resblock_dilations[d], slope=cfg.leaky_slope, device=dev),
                                      cx, None, ox, None))
                        cx, clx = ox, None
                        continue
                    c1_src = cx if in_lds else clx
                    steps.append(("gemm", conv(c1_src, tmp, sd[p + f"convs1.{d}.weight"], sd[p + f"convs1.{d}.bias"],
                                               dilation=cfg.resblock_dilations[d], act="leaky", slope=cfg.leaky_slope,
                                               **(dict(pre_slope=cfg.leaky_slope) if in_lds else {})),
                                  c1_src, None, tmp, None))
                    ol = None if (last or in_lds) else (la if d % 2 == 0 else lb)
                    if reflect:
                        steps.append(("halo", 1, tmp, None, None, None))
                    steps.append(("gemm", conv(tmp, ox, sd[p + f"convs2.{d}.weight"], sd[p + f"convs2.{d}.bias"],
                                               slope2=cfg.leaky_slope), tmp, cx, ox, ol))
                    if reflect and ol is not None:         # the next conv1 reads lrelu(x) reflect-padded (the raw ox only feeds residuals)
                        steps.append(("halo", 1, ol, None, None, None))
                    cx, clx = ox, ol
            nxt = M(t2, co)
            slope = cfg.leaky_slope if i < nstage - 1 else 0.01            # F.leaky_relu default before conv_post
            steps.append(("mix", slope, outs, None, nxt, None))
            if reflect:
                # the mix runs over whole padded maps; the transposed convolution that follows needs a ZERO halo (it is not a
                # "same" conv), conv_post a reflected one
                steps.append(("halo", 1 if i == nstage - 1 else 0, nxt, None, None, None))
            cur, ch, t = nxt, co, t2
        ws = dict(mel=mel, steps=steps, last=cur, T_out=t, wav=torch.empty(B, 1, t, dtype=torch.float32, device=dev))
        ws["flops"] = sum(s[1].flops for s in steps if s[0] == "gemm") + 2.0 * B * t * ch * cfg.post_kernel
        self._ws[key] = ws
        return ws

    def flops(self, B: int, T: int) -> float:
        return self._workspace(B, T)["flops"]

    def decode_batch(self, mel: torch.Tensor) -> torch.Tensor:
        """``mel [B, n_mels, T]`` (fp32, on the GPU) -> ``wav [B, 1, T * hop]`` fp32."""
        if mel.dim() == 2:
            mel = mel[None]
        if mel.dim() != 3 or mel.shape[1] != self.cfg.in_channels:
            raise ValueError(f"mel must be [B, {self.cfg.in_channels}, T]")
        mel = mel.to(self.dev, torch.float32).contiguous()
        B, C, T0 = mel.shape
        pad = self.inference_padding
        T = T0 + 2 * pad
        ws = self._workspace(B, T)
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        sp = self.split
        if sp:
            _lib.check(lib.advh_hifigan_pack_mel_split(mel.data_ptr(), ws["mel"].t.data_ptr(), ws["mel"].t.stride(0), B, C, T0, pad, HALO, st),
                       "advh_hifigan_pack_mel_split")
        elif pad:
            _lib.check(lib.advh_hifigan_pack_mel_pad(mel.data_ptr(), ws["mel"].t.data_ptr(), B, C, T0, pad, HALO, st), "advh_hifigan_pack_mel_pad")
        else:
            _lib.check(lib.advh_hifigan_pack_mel(mel.data_ptr(), ws["mel"].t.data_ptr(), B, C, T, HALO, st), "advh_hifigan_pack_mel")
        for kind, plan, src, resid, dst, dst2 in ws["steps"]:
            if kind == "gemm":
                plan.run(src.t, out_h=dst.t, resid=None if resid is None else resid.t, out_h2=None if dst2 is None else dst2.t)
            elif kind == "halo":                       # split maps: both planes are [B][P][C] images of the same geometry
                _lib.check(lib.advh_halo_fill_f16(src.t.data_ptr(), src.B * (2 if sp else 1), src.T, src.C, src.halo, plan, st), "advh_halo_fill_f16")
            elif sp:
                a, b, c = src
                _lib.check(lib.advh_hifigan_mrf_mix_split(a.t.data_ptr(), b.t.data_ptr(), c.t.data_ptr(), dst.t.data_ptr(), plan,
                                                          dst.t.stride(0), dst.t.stride(0), st), "advh_hifigan_mrf_mix_split")
            else:
                a, b, c = src
                _lib.check(lib.advh_hifigan_mrf_mix(a.t.data_ptr(), b.t.data_ptr(), c.t.data_ptr(), dst.t.data_ptr(), plan,
                                                    dst.t.numel(), st), "advh_hifigan_mrf_mix")
        last = ws["last"]
        if sp:
            _lib.check(lib.advh_hifigan_conv_post_split(last.t.data_ptr(), last.t.stride(0), self.post_w.data_ptr(), self.post_b,
                                                        ws["wav"].data_ptr(), B, last.C, last.T, HALO, self.cfg.post_kernel, st),
                       "advh_hifigan_conv_post_split")
        else:
            _lib.check(lib.advh_hifigan_conv_post(last.t.data_ptr(), self.post_w.data_ptr(), self.post_b, ws["wav"].data_ptr(), B,
                                                  last.C, last.T, HALO, self.cfg.post_kernel, st), "advh_hifigan_conv_post")
        return ws["wav"].clone()
