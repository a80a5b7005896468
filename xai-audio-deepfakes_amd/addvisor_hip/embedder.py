"""Frozen wav2vec2 embedder + logreg head on the HIP kernels.

Host-side orchestration of ``AudioProcessor.extract_features`` (audioprocessor.py:69-77) and the
pool + ``TorchLogReg`` head (classifier_embedder.py:21-38, LMAC_metrics.py:130): the arithmetic of
HF ``Wav2Vec2Model.forward`` (transformers/models/wav2vec2/modeling_wav2vec2.py:254-802, eval mode)
is issued as a fixed sequence of C-ABI kernel launches on the current stream.  Nothing here computes
on the host, and there is no fallback: every op is a hand-written gfx950 kernel.

Precision (``precision=``):
  "f16"  GEMM operands and the activations between GEMMs are fp16, accumulation is fp32, the residual stream /
         LayerNorm / softmax statistics are fp32;
  "f32"  the fp32-class mode: the same launches on split-format plane pairs (x = hi + lo * 2^-11, ~22 bits; three MFMAs
         per product, csrc/device_math.h) -- the arithmetic class of the reference, which runs the embedder in fp32
         (audioprocessor.py:69-77).  The line-tile positional conv and the fp16 branch fusion are fp16-only and are
         replaced by the implicit GEMM / fp32 residual epilogues.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, gemm as G
from .synthetic import EmbedderConfig

FE_SLACK_ROWS = 4
BRANCH_F16 = os.environ.get("ADDVISOR_BRANCH_F16", "1") != "0"        # post-LN layers: residual add fused into the LayerNorm
POSCONV_TILE = os.environ.get("ADDVISOR_POSCONV_TILE", "1") != "0"     # A/B switch: 0 = implicit GEMM for the positional conv


class PosconvDesc(ctypes.Structure):
    """advh_posconv_desc (include/addvisor_hip.h)."""
    _fields_ = [("xg", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("resid", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("B", ctypes.c_int), ("T", ctypes.c_int), ("H", ctypes.c_int),
                ("G", ctypes.c_int), ("K", ctypes.c_int)]


def _f32(t, dev):
    return t.detach().to(torch.float32).contiguous().to(dev)


def default_precision() -> str:
    """``ADDVISOR_PRECISION`` = f32 (default: the reference's arithmetic class) | f16 (fp16 operands, 2-3x faster)."""
    p = os.environ.get("ADDVISOR_PRECISION", "f32").lower()
    if p not in ("f16", "f32"):
        raise ValueError("ADDVISOR_PRECISION must be f16 or f32")
    return p


class _LN:
    def __init__(self, sd, prefix, dev):
        self.g, self.b = _f32(sd[prefix + ".weight"], dev), _f32(sd[prefix + ".bias"], dev)
        self.C = self.g.numel()

    def __call__(self, x: torch.Tensor, M: int, eps: float, out_f=None, out_h=None, gelu=False, add_h=None, split=False):
        """``LayerNorm(x [+ add_h])``; ``add_h``: fp16 [M, C] branch output added to the fp32 residual rows first.
        ``split``: fp16 tensors are ``[2, ...]`` plane pairs (fp32-class mode)."""
        C = self.C
        if split:
            in_f32 = x.dtype == torch.float32
            assert add_h is None
            _lib.check(_lib.lib().advh_layernorm_split(
                x.data_ptr(), int(in_f32), C, 0 if in_f32 else x.stride(0), None, C, 0, self.g.data_ptr(), self.b.data_ptr(),
                None if out_f is None else out_f.data_ptr(), None if out_h is None else out_h.data_ptr(), C,
                0 if out_h is None else out_h.stride(0), M, C, eps, int(gelu), torch.cuda.current_stream().cuda_stream),
                "advh_layernorm_split")
            return
        _lib.check(_lib.lib().advh_layernorm_add(
            x.data_ptr(), int(x.dtype == torch.float32), C, None if add_h is None else add_h.data_ptr(), C,
            self.g.data_ptr(), self.b.data_ptr(),
            None if out_f is None else out_f.data_ptr(), None if out_h is None else out_h.data_ptr(), C, M, C,
            eps, int(gelu), torch.cuda.current_stream().cuda_stream), "advh_layernorm_add")


class HipEmbedder:
    """``forward(wave[B, n]) -> hidden_states[layer_index] [B,T,H] fp32, logits [B,1], probs [B,1]``."""

    def __init__(self, cfg: EmbedderConfig, sd: Dict[str, torch.Tensor], coef, intercept, device, precision: Optional[str] = None):
        _lib.init()
        self.cfg, self.dev = cfg, device
        self.precision = precision or default_precision()
        if self.precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.split = self.precision == "f32"
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        if cfg.conv_kernel[0] != 10 or cfg.conv_stride[0] != 5:
            raise ValueError("feature-encoder layer 0 must be Conv1d(k=10, stride=5)")
        if cfg.head_dim % 8 or cfg.head_dim > 128:
            raise ValueError("attention kernel supports head dims that are multiples of 8 up to 128")
        self.nl = min(cfg.layer_index, cfg.num_hidden_layers)
        dev = device
        p0 = "feature_extractor.conv_layers.0."
        self.w0 = _f32(self.sd[p0 + "conv.weight"].reshape(cfg.conv_dim[0], 10), dev)
        self.b0 = _f32(self.sd[p0 + "conv.bias"], dev) if cfg.conv_bias else None
        self.layer_mode = cfg.feat_extract_norm == "layer"
        self.fe_ln: List[Optional[_LN]] = []
        for i in range(len(cfg.conv_dim)):
            has = self.layer_mode or i == 0
            self.fe_ln.append(_LN(self.sd, f"feature_extractor.conv_layers.{i}.layer_norm", dev) if has else None)
        self.fp_ln = _LN(self.sd, "feature_projection.layer_norm", dev)
        self.enc_ln = _LN(self.sd, "encoder.layer_norm", dev)
        self.ln1 = [_LN(self.sd, f"encoder.layers.{l}.layer_norm", dev) for l in range(self.nl)]
        self.ln2 = [_LN(self.sd, f"encoder.layers.{l}.final_layer_norm", dev) for l in range(self.nl)]
        self.coef = _f32(torch.as_tensor(np.asarray(coef)).reshape(-1), dev)
        self.intercept = float(np.asarray(intercept).reshape(-1)[0])
        if self.coef.numel() != cfg.hidden_size:
            raise ValueError("logreg coef_ must have hidden_size entries")
        self._ws: Dict[Tuple[int, int, int], dict] = {}
        self._wcache: dict = {}                    # packed fp16 weights, shared by the plans of every batch shape

    def f16_twin(self) -> "HipEmbedder":
        """The fp16-operand instance of the same model: the input-gradient chain (``EmbedderGrad``: Saliency / IG / LMAC
        loss backward) runs on the fp16 kernels and shares the forward plans of an fp16 embedder."""
        if not self.split:
            return self
        if getattr(self, "_twin", None) is None:
            self._twin = HipEmbedder(self.cfg, self.sd, self.coef.cpu().numpy(), self.intercept, self.dev, precision="f16")
        return self._twin

    # ------------------------------------------------------------------ planning (once per batch shape)
    def _lengths(self, L: int) -> List[int]:
        out, n = [], L
        for k, s in zip(self.cfg.conv_kernel, self.cfg.conv_stride):
            n = (n - k) // s + 1
            out.append(n)
        return out

    def _workspace(self, B: int, L: int, slot: int = 0) -> dict:
        key = (B, L, slot)
        if key in self._ws:
            return self._ws[key]
        cfg, dev, sd = self.cfg, self.dev, self.sd
        Ls = self._lengths(L)
        nfe = len(Ls)
        strides = cfg.conv_stride
        # padded row counts: P[i] = P[i+1] * stride[i+1], every P[i] >= L[i]
        P_last = 1
        for i in range(nfe):
            prod = 1
            for j in range(i + 1, nfe):
                prod *= strides[j]
            P_last = max(P_last, -(-(Ls[i] + 1) // prod))      # >= 1 zero filler row per clip (the backward relies on it)
        P = [0] * nfe
        P[-1] = P_last
        for i in range(nfe - 2, -1, -1):
            P[i] = P[i + 1] * strides[i + 1]
        T, H, I = Ls[-1], cfg.hidden_size, cfg.intermediate_size
        M = B * T
        C = cfg.conv_dim
        h16, f32 = torch.float16, torch.float32
        sp = self.split
        pl = (2,) if sp else ()                      # leading plane dimension of every fp16 tensor in the fp32-class mode
        ws = dict(B=B, L=L, Ls=Ls, P=P, T=T, M=M)
        ws["stats"] = torch.empty(B, 2, dtype=f32, device=dev)
        ws["norm"] = torch.empty(B, C[0], 2, dtype=f32, device=dev)
        ws["mr"] = torch.empty(B, C[0], 2, dtype=f32, device=dev)
        # + FE_SLACK_ROWS readable rows behind each buffer: the last filler row of a layer reaches k - stride rows past it
        ws["fe"] = [torch.zeros(pl + (B * P[0] * C[0] + FE_SLACK_ROWS * max(C),), dtype=h16, device=dev),
                    torch.zeros(pl + (B * P[1] * C[1] + FE_SLACK_ROWS * max(C),), dtype=h16, device=dev)]
        ws["feat"] = torch.empty(pl + (M, C[-1]), dtype=h16, device=dev)
        ws["featn"] = torch.empty(pl + (M, C[-1]), dtype=h16, device=dev)
        ws["h"] = torch.empty(M, H, dtype=f32, device=dev)
        ws["h16"] = torch.empty(pl + (M, H), dtype=h16, device=dev)
        ws["qkv"] = torch.empty(pl + (M, 3 * H), dtype=h16, device=dev)
        ws["ctx"] = torch.empty(pl + (M, H), dtype=h16, device=dev)
        ws["br"] = None if sp else torch.empty(M, H, dtype=h16, device=dev)   # fp16 branch output (attention / feed-forward projection)
        ws["ffn"] = torch.empty(pl + (M, I), dtype=h16, device=dev)
        K, Gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        Cg = H // Gp
        ws["xg"] = torch.empty(pl + (Gp, B, T + K, Cg), dtype=h16, device=dev)
        ws["logit"] = torch.empty(B, dtype=f32, device=dev)
        ws["prob"] = torch.empty(B, dtype=f32, device=dev)

        act = "none" if self.layer_mode else "gelu"
        fe_plans = []
        for i in range(1, nfe):
            p = f"feature_extractor.conv_layers.{i}.conv."
            fe_plans.append(G.plan_conv1d_cl(B, P[i - 1], P[i], Ls[i], sd[p + "weight"], sd.get(p + "bias"),
                                             strides[i], act=act, compact_out=(i == nfe - 1), device=dev,
                                             cache=(self._wcache, ("fe", i)), slack_rows=FE_SLACK_ROWS, split=sp))
        ws["fe_plans"] = fe_plans
        ws["proj"] = G.plan_linear(M, sd["feature_projection.projection.weight"],
                                   sd["feature_projection.projection.bias"], device=dev, cache=(self._wcache, "proj"), split=sp)
        # positional conv: weight_norm folded (modeling_wav2vec2.py:326-357), one GEMM batched over groups
        g0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
        v0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
        w2 = lambda: (g0 * v0 / v0.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()).view(Gp, Cg, Cg, K) \
            .permute(0, 1, 3, 2).reshape(Gp, Cg, K * Cg)                            # [g][n][(k, ci)]
        cc = Cg // 8
        ws["pos"] = G.GemmPlan(M=M, N=Cg, w2=w2, ktab=np.arange(K * cc, dtype=np.int64),
                               sources=[G.Source((T + K) * cc, 0, cc, 0, sZ=B * (T + K) * cc)], Hg=1, Wg=T,
                               window=(0, 1, 0, T), halo_zero=False, out=(T * H, 0, H, 0), n_div=G.round_up(Cg, 4),
                               o_sZ=Cg, nz=Gp, bias=sd["encoder.pos_conv_embed.conv.bias"], bias_sZ=Cg, act="gelu",
                               device=dev, cache=(self._wcache, "pos"), split=sp)
        # line-tile launch for the same layer (csrc/posconv_tile.hip): the clip's gathered rows staged in LDS once, the
        # group's weights streamed; the implicit GEMM above stays as the fallback for other geometries
        ws["pos_tile"] = None
        # (64-channel groups leave room for one workgroup per CU only: measured 308 vs 369 us at 64 clips but 925 vs 876 us
        # at 192 -- tools/bench_posconv.py -- so large models switch back to the GEMM for big batches)
        if POSCONV_TILE and not sp and _lib.lib().advh_posconv_tile_lds_bytes(Cg, T) > 0 and K == 128 and (Cg <= 48 or B <= 96):
            if "pos_tile" not in self._wcache:
                wt = w2().view(Gp, Cg, K * Cg // 32, 32).permute(0, 2, 1, 3).contiguous().to(torch.float16)   # [g][k-step][n][32]
                self._wcache["pos_tile"] = (wt.to(dev), sd["encoder.pos_conv_embed.conv.bias"].float().contiguous().to(dev))
            d = PosconvDesc()
            d.B, d.T, d.H, d.G, d.K = B, T, H, Gp, K
            d.W, d.bias = self._wcache["pos_tile"][0].data_ptr(), self._wcache["pos_tile"][1].data_ptr()
            ws["pos_tile"] = d
        layers = []
        for l in range(self.nl):
            p = f"encoder.layers.{l}."
            if ("qkv", l) in self._wcache:                 # packed already: only the shapes are needed
                wqkv = torch.empty(3 * H, H, device="meta")
                bqkv = torch.empty(3 * H, device="meta")
            else:
                wqkv = torch.cat([sd[p + f"attention.{n}_proj.weight"] for n in ("q", "k", "v")], 0)
                bqkv = torch.cat([sd[p + f"attention.{n}_proj.bias"] for n in ("q", "k", "v")], 0)
            c = lambda name: (self._wcache, (name, l))
            layers.append(dict(
                qkv=G.plan_linear(M, wqkv, bqkv, device=dev, cache=c("qkv"), split=sp),
                out=G.plan_linear(M, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"], device=dev,
                                  cache=c("out"), split=sp),
                ff1=G.plan_linear(M, sd[p + "feed_forward.intermediate_dense.weight"],
                                  sd[p + "feed_forward.intermediate_dense.bias"], act="gelu", device=dev, cache=c("ff1"), split=sp),
                ff2=G.plan_linear(M, sd[p + "feed_forward.output_dense.weight"],
                                  sd[p + "feed_forward.output_dense.bias"], device=dev, cache=c("ff2"), split=sp)))
        ws["layers"] = layers
        ws["flops"] = (sum(p.flops for p in fe_plans) + ws["proj"].flops + ws["pos"].flops
                       + sum(sum(pl.flops for pl in lay.values()) for lay in layers)
                       + self.nl * 4.0 * B * T * T * H + 2.0 * B * Ls[0] * C[0] * 10)
        self._ws[key] = ws
        return ws

    def flops(self, B: int, L: int) -> float:
        """Algorithmic FLOPs (2*MAC) of one forward for B clips of L samples."""
        return self._workspace(B, L)["flops"]

    # ------------------------------------------------------------------ forward
    def forward(self, wave: torch.Tensor, length: Optional[int] = None, want_hidden: bool = True,
                normalize: bool = True, slot: int = 0):
        """Returns ``(hidden [B,T,H] fp32 or None, logits [B,1], probs [B,1])`` -- fresh tensors."""
        if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_cuda:
            raise ValueError("wave must be a CUDA fp32 tensor [B, n]")
        wave = wave.contiguous()
        B, n_in = wave.shape
        L = n_in if length is None else int(length)
        cfg, lib = self.cfg, _lib.lib()
        ws = self._workspace(B, L, slot)           # `slot` gives concurrent streams their own buffers
        st = torch.cuda.current_stream().cuda_stream
        Ls, P, T, M, H = ws["Ls"], ws["P"], ws["T"], ws["M"], cfg.hidden_size
        eps = cfg.layer_norm_eps
        C = cfg.conv_dim
        a, bbuf = ws["fe"]
        mode = 1 if self.layer_mode else 0
        ln0 = self.fe_ln[0]
        sp = self.split
        if sp:
            _lib.check(lib.advh_w2v2_frontend_split(
                wave.data_ptr(), wave.stride(0), n_in, B, L, self.w0.data_ptr(),
                None if self.b0 is None else self.b0.data_ptr(), ln0.g.data_ptr(), ln0.b.data_ptr(), mode, int(normalize),
                ws["stats"].data_ptr(), ws["norm"].data_ptr(), ws["mr"].data_ptr(), a.data_ptr(), a.stride(0), Ls[0], P[0], C[0], st),
                "advh_w2v2_frontend_split")
        else:
            _lib.check(lib.advh_w2v2_frontend(
                wave.data_ptr(), wave.stride(0), n_in, B, L, self.w0.data_ptr(),
                None if self.b0 is None else self.b0.data_ptr(), ln0.g.data_ptr(), ln0.b.data_ptr(), mode, int(normalize),
                ws["stats"].data_ptr(), ws["norm"].data_ptr(), ws["mr"].data_ptr(), a.data_ptr(), Ls[0], P[0], C[0], st),
                "advh_w2v2_frontend")
        if self.layer_mode:
            ln0(a, B * P[0], 1e-5, out_h=a, gelu=True, split=sp)
        cur, nxt = a, bbuf
        nfe = len(Ls)
        for i in range(1, nfe):
            last = i == nfe - 1
            dst = ws["feat"] if last else nxt
            ws["fe_plans"][i - 1].run(cur, out_h=dst)
            if self.layer_mode:
                self.fe_ln[i](dst, M if last else B * P[i], 1e-5, out_h=dst, gelu=True, split=sp)
            cur, nxt = dst, cur
        h, h16 = ws["h"], ws["h16"]
        self.fp_ln(ws["feat"], M, eps, out_h=ws["featn"], split=sp)
        ws["proj"].run(ws["featn"], out_f=h)
        K, Gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        if sp:
            _lib.check(lib.advh_posconv_gather_split(h.data_ptr(), ws["xg"].data_ptr(), ws["xg"].stride(0), B, T, H, Gp, K, K // 2, st),
                       "advh_posconv_gather_split")
        else:
            _lib.check(lib.advh_posconv_gather(h.data_ptr(), ws["xg"].data_ptr(), B, T, H, Gp, K, K // 2, None, st), "advh_posconv_gather")
        if ws["pos_tile"] is not None:                                # h += gelu(pos_conv(h))
            d = ws["pos_tile"]
            d.xg, d.resid, d.out = ws["xg"].data_ptr(), h.data_ptr(), h.data_ptr()
            _lib.check(lib.advh_posconv_tile_f16(ctypes.byref(d), st), "advh_posconv_tile_f16")
        else:
            ws["pos"].run(ws["xg"], out_f=h, resid=h)
        stable = cfg.do_stable_layer_norm
        if not stable:
            self.enc_ln(h, M, eps, out_f=h, out_h=h16, split=sp)
        for l in range(self.nl):
            lay = ws["layers"][l]
            if stable:
                self.ln1[l](h, M, eps, out_h=h16, split=sp)
            lay["qkv"].run(h16, out_h=ws["qkv"])
            if sp:
                _lib.check(lib.advh_attention_split(ws["qkv"].data_ptr(), ws["qkv"].stride(0), ws["ctx"].data_ptr(), ws["ctx"].stride(0),
                                                    B, T, H, cfg.num_attention_heads, st), "advh_attention_split")
            else:
                _lib.check(lib.advh_attention_f16(ws["qkv"].data_ptr(), ws["ctx"].data_ptr(), B, T, H,
                                                  cfg.num_attention_heads, st), "advh_attention_f16")
            if not stable and BRANCH_F16 and not sp:
                # post-LN layer: the projection stores its fp16 result only, the residual add happens inside the LayerNorm
                # (fp32 stream + fp16 branch): the GEMM epilogue neither reads nor re-writes the fp32 rows
                lay["out"].run(ws["ctx"], out_h=ws["br"])
                self.ln1[l](h, M, eps, out_f=h, out_h=h16, add_h=ws["br"])
                lay["ff1"].run(h16, out_h=ws["ffn"])
                lay["ff2"].run(ws["ffn"], out_h=ws["br"])
                self.ln2[l](h, M, eps, out_f=h, out_h=h16, add_h=ws["br"])
                continue
            lay["out"].run(ws["ctx"], out_f=h, resid=h)               # h = h + out_proj(ctx)
            if stable:
                self.ln2[l](h, M, eps, out_h=h16, split=sp)
            else:
                self.ln1[l](h, M, eps, out_f=h, out_h=h16, split=sp)
            lay["ff1"].run(h16, out_h=ws["ffn"])
            lay["ff2"].run(ws["ffn"], out_f=h, resid=h)               # h = h + ffn(h)
            if not stable:
                self.ln2[l](h, M, eps, out_f=h, out_h=h16, split=sp)
        if stable and self.nl == cfg.num_hidden_layers:
            self.enc_ln(h, M, eps, out_f=h)
        _lib.check(lib.advh_pool_logreg(h.data_ptr(), self.coef.data_ptr(), self.intercept, ws["logit"].data_ptr(),
                                        ws["prob"].data_ptr(), None, B, T, H, st), "advh_pool_logreg")
        hid = h.view(B, T, H).clone() if want_hidden else None
        return hid, ws["logit"].clone().view(B, 1), ws["prob"].clone().view(B, 1)
