"""Input gradient of the frozen classifier  F(x) = logreg(mean_t(hidden_states[k](wav2vec2(norm(x)))))
on the HIP kernels: the quantity Captum's Saliency / InputXGradient / IntegratedGradients differentiate
(captum_saliency.py:84-100, 116-135).

The weights are frozen, so the backward pass is dgrad-only: every dense product is the forward's implicit GEMM
with the transposed weight (``gemm.plan_linear(W.T)``, ``plan_conv1d_dgrad``), with the activation derivative
(``dact_src``) fused in its epilogue; LayerNorm, attention and the waveform front end have dedicated backward
kernels (csrc/backward.hip, frontend_bwd.hip).  The forward of this class is the same arithmetic as
``HipEmbedder.forward`` but writes every tensor the backward needs (pre-activations, LayerNorm inputs, QKV)
to its own buffer instead of updating in place.

Precision follows the embedder (``precision=None``): with an fp32-class embedder ("f32", the reference's arithmetic class --
captum_saliency.py:116-135 and loss_function.py:46-53 differentiate with fp32 autograd) every saved activation and every
gradient between GEMMs is a split-format plane pair (hi + lo * 2^-11, ~22 bits; csrc/device_math.h), the dgrad GEMMs run
three fp16 MFMAs per product on transposed split weights, LayerNorm / front-end backward read and write plane pairs and the
attention backward runs on the fp32-input matrix instruction (csrc/attention_bwd_f32.hip).  ``precision="f16"`` keeps the
fp16-operand chain (fp16 gradients between GEMMs).  In both modes the gradients are multiplied by ``loss_scale`` (a power of
two: exact, the chain is linear in the gradient) so that small values stay in the fp16 exponent range of the planes, and
the residual-stream gradient is fp32.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, gemm as G
from .embedder import FE_SLACK_ROWS, HipEmbedder


def plan_conv1d_dgrad(B: int, P_out: int, weight: torch.Tensor, stride: int, device=None, cache=None, split: bool = False) -> G.GemmPlan:
    """Input gradient of the channels-last Conv1d of ``gemm.plan_conv1d_cl`` (no padding): position
    ``u = s*q + phase`` of the input receives ``sum_{i} dZ[q - (nt-1) + i] . W[:, :, phase + s*(nt-1-i)]``
    (nt = ceil(k/s) taps), so the layer is a GEMM over rows q with K = nt*Cout (nt adjacent dZ rows = one
    contiguous slab) and N = s*Cin, whose row q of the output is input rows s*q .. s*q+s-1.
    ``A0`` must point ``nt-1`` rows BEFORE dZ[0] (a zero guard row), and every clip needs >= 1 zero filler row."""
    Cout, Cin, k = weight.shape
    s = stride
    nt = -(-k // s)
    assert Cout % 8 == 0 and Cin % 4 == 0
    def w2():
        m = torch.zeros(s * Cin, nt * Cout)
        for phase in range(s):
            for i in range(nt):
                j = phase + s * (nt - 1 - i)
                if j < k:
                    m[phase * Cin:(phase + 1) * Cin, i * Cout:(i + 1) * Cout] = weight[:, :, j].t()
        return m[None]
    cc = Cout // 8
    return G.GemmPlan(M=B * P_out, N=s * Cin, w2=w2, ktab=np.arange(nt * cc, dtype=np.int64),
                      sources=[G.Source(P_out * cc, 0, cc, 0)], Hg=1, Wg=P_out, window=(0, 1, 0, P_out), halo_zero=False,
                      out=(P_out * s * Cin, 0, s * Cin, 0), device=device, cache=cache, split=split), nt


class EmbedderGrad:
    def __init__(self, emb: HipEmbedder, precision: Optional[str] = None):
        """``precision``: None = the embedder's own ("f32": split-format chain; "f16": fp16 chain), or "f16" to run the fp16
        chain next to an fp32-class embedder (``HipEmbedder.f16_twin``)."""
        if precision not in (None, "f16", "f32"):
            raise ValueError("precision must be None, 'f16' or 'f32'")
        if precision == "f16":
            emb = emb.f16_twin()
        elif precision == "f32" and not emb.split:
            raise ValueError("an fp32-class gradient chain needs an fp32-class embedder (HipEmbedder(precision='f32'))")
        self.emb = emb
        self.split = emb.split
        self.precision = emb.precision
        self.cfg, self.dev, self.sd = emb.cfg, emb.dev, emb.sd
        self.layer_mode = emb.layer_mode                       # "layer" feature extractor (wav2vec2-large / xls-r)
        self.stable = self.cfg.do_stable_layer_norm            # pre-LN encoder
        self._ws: Dict[Tuple[int, int], dict] = {}
        self._wcache: dict = {}

    # ------------------------------------------------------------------ buffers and plans
    def _workspace(self, B: int, L: int) -> dict:
        key = (B, L)
        if key in self._ws:
            return self._ws[key]
        emb, cfg, dev, sd, sp = self.emb, self.cfg, self.dev, self.sd, self.split
        f = emb._workspace(B, L)                      # forward plans / shapes are shared
        Ls, P, T, M = f["Ls"], f["P"], f["T"], f["M"]
        for i in range(len(Ls)):
            if P[i] <= Ls[i]:
                raise ValueError("backward needs one zero filler row per clip (P > L)")
        H, I, C = cfg.hidden_size, cfg.intermediate_size, cfg.conv_dim
        nfe, nl = len(Ls), emb.nl
        h16, f32 = torch.float16, torch.float32
        pl = (2,) if sp else ()                       # fp32-class chain: every fp16 tensor is a [2, ...] plane pair
        z = lambda *s, dt=h16: torch.zeros(*((pl + tuple(s)) if dt == h16 else s), dtype=dt, device=dev)
        w = dict(f=f)
        # Feature-encoder level i (i < nfe - 1): y_i (post-GELU), z_i (pre-activation) and dZ_i all live in buffers of ONE
        # shape -- a zero guard row, the [B][P_i][C_i] rows, then the readable slack rows the affine-row loader of the next
        # layer's plan may touch (gemm.plan_conv1d_cl(slack_rows=...)) -- so that in the fp32-class mode the fp16-side operands
        # of one GEMM epilogue (out_h, out_pre, dact_src) share one plane pitch (advh_gemm_desc.o_lo).
        lvl = lambda i: z((1 + B * P[i] + FE_SLACK_ROWS) * C[i])
        self._body = lambda t, i: t[..., C[i]:]                                   # rows from the first data row on (guard skipped)
        # forward saves ------------------------------------------------------------------
        w["y"] = [lvl(i) for i in range(nfe - 1)]                                  # post-GELU outputs of layers 0..nfe-2
        w["z"] = [lvl(0) if self.layer_mode else None]                             # pre-norm / pre-GELU outputs of the convs
        w["z"] += [lvl(i) for i in range(1, nfe - 1)] + [z(M, C[-1])]
        w["dyb"] = lvl(0) if self.layer_mode else None
        w["dfeat"] = z(M, C[-1]) if self.layer_mode else None
        w["t16"] = z(M, H)
        w["xf"] = z(M, H, dt=f32)
        w["feat"] = z(M, C[-1])
        w["pc"] = z(M, H)                                                        # pre-GELU positional conv
        w["h1"] = z(M, H, dt=f32)                                                # input of encoder.layer_norm
        w["x"] = [z(M, H, dt=f32) for _ in range(nl + 1)]                        # layer inputs / final output
        w["qkv"] = [z(M, 3 * H) for _ in range(nl)]
        w["s1"] = [z(M, H, dt=f32) for _ in range(nl)]
        w["m"] = [z(M, H, dt=f32) for _ in range(nl)]
        w["g1"] = [z(M, I) for _ in range(nl)]
        w["s2"] = [z(M, H, dt=f32) for _ in range(nl)]
        # backward scratch ---------------------------------------------------------------
        w["dlogit"] = z(B, dt=f32)
        w["da"] = z(M, H, dt=f32)
        w["db"] = z(M, H, dt=f32)
        w["d16"] = z(M, H)
        w["dI"] = z(M, I)
        w["dctx"] = z(M, H)
        w["dqkv"] = z(M, 3 * H)
        w["dfeatn"] = z(M, C[-1])
        # dZ_i: per-clip padded layout [B][P_i][C_i] with one zero guard row in front
        w["dz"] = [lvl(i) for i in range(nfe - 1)] + [z((B * P[-1] + 1) * C[-1])]
        w["g"] = z(B * P[0], 16, dt=f32)
        ntile = -(-P[0] // 64)
        w["part"] = z(B, ntile, C[0], 2, dt=f32)
        w["sums"] = z(B, C[0], 2, dt=f32)
        w["dxh"] = z(B, L, dt=f32)
        w["wpart"] = z(B, -(-L // 2048), 2, dt=f32)
        # backward plans -----------------------------------------------------------------
        wc = self._wcache
        lin = lambda wt, key: G.plan_linear(M, wt, None, device=dev, cache=(wc, key), split=sp)
        layers = []
        for l in range(nl):
            p = f"encoder.layers.{l}."
            if ("qkv", l) in wc:
                wqkv = torch.empty(3 * H, H, device="meta")
            else:
                wqkv = torch.cat([sd[p + f"attention.{n}_proj.weight"] for n in ("q", "k", "v")], 0)
            layers.append(dict(ff2=lin(sd[p + "feed_forward.output_dense.weight"].t(), ("ff2", l)),
                               ff1=lin(sd[p + "feed_forward.intermediate_dense.weight"].t(), ("ff1", l)),
                               out=lin(sd[p + "attention.out_proj.weight"].t(), ("out", l)), qkv=lin(wqkv.t(), ("qkv", l))))
        w["layers"] = layers
        w["proj"] = lin(sd["feature_projection.projection.weight"].t(), "proj")
        K, Gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        Cg, cc = H // Gp, H // Gp // 8
        g0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
        v0 = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
        w2b = lambda: (g0 * v0 / v0.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()).view(Gp, Cg, Cg, K) \
            .flip(3).permute(0, 2, 3, 1).reshape(Gp, Cg, K * Cg)                   # [g][ci][(k', co)], taps flipped
        w["pos"] = G.GemmPlan(M=M, N=Cg, w2=w2b, ktab=np.arange(K * cc, dtype=np.int64),
                              sources=[G.Source((T + K) * cc, 0, cc, 0, sZ=B * (T + K) * cc)], Hg=1, Wg=T,
                              window=(0, 1, 0, T), halo_zero=False, out=(T * H, 0, H, 0), n_div=G.round_up(Cg, 4),
                              o_sZ=Cg, nz=Gp, device=dev, cache=(wc, "pos"), split=sp)
        fe = []
        for i in range(1, nfe):
            plan, nt = plan_conv1d_dgrad(B, P[i], sd[f"feature_extractor.conv_layers.{i}.conv.weight"], cfg.conv_stride[i], dev,
                                         cache=(wc, ("fe", i)), split=sp)
            fe.append((plan, nt))
        w["fe"] = fe
        w0t = torch.zeros(16, C[0])
        w0t[:10] = sd["feature_extractor.conv_layers.0.conv.weight"].reshape(C[0], 10).t()
        w["g_plan"] = G.plan_linear(B * P[0], w0t, None, device=dev, cache=(wc, "w0t"), split=sp)
        self._ws[key] = w
        return w

    # ------------------------------------------------------------------ forward with saves
    def forward(self, wave: torch.Tensor, length: Optional[int] = None):
        emb, cfg, lib, sp = self.emb, self.cfg, _lib.lib(), self.split
        wave = wave.contiguous()
        B, n_in = wave.shape
        L = n_in if length is None else int(length)
        w = self._workspace(B, L)
        f = w["f"]
        st = torch.cuda.current_stream().cuda_stream
        Ls, P, T, M, H = f["Ls"], f["P"], f["T"], f["M"], cfg.hidden_size
        eps, C, nfe, nl = cfg.layer_norm_eps, cfg.conv_dim, len(f["Ls"]), emb.nl
        ln0 = emb.fe_ln[0]
        lm = self.layer_mode
        body = self._body
        y = [body(t, i) for i, t in enumerate(w["y"])]
        zb = [None if t is None else (body(t, i) if i < nfe - 1 else t) for i, t in enumerate(w["z"])]
        out0 = zb[0] if lm else y[0]
        if sp:
            _lib.check(lib.advh_w2v2_frontend_split(
                wave.data_ptr(), wave.stride(0), n_in, B, L, emb.w0.data_ptr(), None if emb.b0 is None else emb.b0.data_ptr(),
                ln0.g.data_ptr(), ln0.b.data_ptr(), 1 if lm else 0, 1, f["stats"].data_ptr(), f["norm"].data_ptr(), f["mr"].data_ptr(),
                out0.data_ptr(), out0.stride(0), Ls[0], P[0], C[0], st), "advh_w2v2_frontend_split")
        else:
            _lib.check(lib.advh_w2v2_frontend(
                wave.data_ptr(), wave.stride(0), n_in, B, L, emb.w0.data_ptr(), None if emb.b0 is None else emb.b0.data_ptr(),
                ln0.g.data_ptr(), ln0.b.data_ptr(), 1 if lm else 0, 1, f["stats"].data_ptr(), f["norm"].data_ptr(), f["mr"].data_ptr(),
                out0.data_ptr(), Ls[0], P[0], C[0], st), "advh_w2v2_frontend")
        if lm:
            ln0(zb[0], B * P[0], 1e-5, out_h=y[0], gelu=True, split=sp)
        for i in range(1, nfe):
            last = i == nfe - 1
            dst = w["feat"] if last else y[i]
            if lm:                                         # conv (+bias) -> z_i ; LayerNorm + GELU -> y_i
                f["fe_plans"][i - 1].run(y[i - 1], out_h=zb[i])
                emb.fe_ln[i](zb[i], M if last else B * P[i], 1e-5, out_h=dst, gelu=True, split=sp)
            else:
                f["fe_plans"][i - 1].run(y[i - 1], out_h=dst, out_pre=zb[i])
        emb.fp_ln(w["feat"], M, eps, out_h=f["featn"], split=sp)
        h = f["h"]
        f["proj"].run(f["featn"], out_f=h)
        K, Gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        if sp:
            _lib.check(lib.advh_posconv_gather_split(h.data_ptr(), f["xg"].data_ptr(), f["xg"].stride(0), B, T, H, Gp, K, K // 2, st),
                       "advh_posconv_gather_split")
        else:
            _lib.check(lib.advh_posconv_gather(h.data_ptr(), f["xg"].data_ptr(), B, T, H, Gp, K, K // 2, None, st), "advh_posconv_gather")
        h16 = f["h16"]
        if self.stable:
            f["pos"].run(f["xg"], out_f=w["x"][0], resid=h, out_pre=w["pc"])
        else:
            f["pos"].run(f["xg"], out_f=w["h1"], resid=h, out_pre=w["pc"])
            emb.enc_ln(w["h1"], M, eps, out_f=w["x"][0], out_h=h16, split=sp)
        for l in range(nl):
            lay = f["layers"][l]
            if self.stable:
                emb.ln1[l](w["x"][l], M, eps, out_h=h16, split=sp)
            lay["qkv"].run(h16, out_h=w["qkv"][l])
            if sp:
                _lib.check(lib.advh_attention_split(w["qkv"][l].data_ptr(), w["qkv"][l].stride(0), f["ctx"].data_ptr(), f["ctx"].stride(0),
                                                    B, T, H, cfg.num_attention_heads, st), "advh_attention_split")
            else:
                _lib.check(lib.advh_attention_f16(w["qkv"][l].data_ptr(), f["ctx"].data_ptr(), B, T, H, cfg.num_attention_heads, st),
                           "advh_attention_f16")
            if self.stable:                                # x_{l+1} = m + ffn(LN2(m)),  m = x_l + attn(LN1(x_l))
                lay["out"].run(f["ctx"], out_f=w["m"][l], resid=w["x"][l])
                emb.ln2[l](w["m"][l], M, eps, out_h=h16, split=sp)
                lay["ff1"].run(h16, out_h=f["ffn"], out_pre=w["g1"][l])
                lay["ff2"].run(f["ffn"], out_f=w["x"][l + 1], resid=w["m"][l])
            else:                                          # x_{l+1} = LN2(m + ffn(m)),  m = LN1(x_l + attn(x_l))
                lay["out"].run(f["ctx"], out_f=w["s1"][l], resid=w["x"][l])
                emb.ln1[l](w["s1"][l], M, eps, out_f=w["m"][l], out_h=h16, split=sp)
                lay["ff1"].run(h16, out_h=f["ffn"], out_pre=w["g1"][l])
                lay["ff2"].run(f["ffn"], out_f=w["s2"][l], resid=w["m"][l])
                emb.ln2[l](w["s2"][l], M, eps, out_f=w["x"][l + 1], out_h=h16, split=sp)
        final = w["x"][nl]
        self._final_ln = self.stable and nl == cfg.num_hidden_layers
        if self._final_ln:
            emb.enc_ln(w["x"][nl], M, eps, out_f=w["xf"], split=sp)
            final = w["xf"]
        _lib.check(lib.advh_pool_logreg(final.data_ptr(), emb.coef.data_ptr(), emb.intercept, f["logit"].data_ptr(),
                                        f["prob"].data_ptr(), None, B, T, H, st), "advh_pool_logreg")
        self._last = (wave, B, n_in, L)
        return f["logit"].clone().view(B, 1), f["prob"].clone().view(B, 1)

    # ------------------------------------------------------------------ backward
    def _ln_bwd(self, ln, x, dy, M, out_f=None, out_h=None, add=None, dact=None, remap=(0, 0), gelu=False, eps=None):
        eps = self.cfg.layer_norm_eps if eps is None else eps
        st = torch.cuda.current_stream().cuda_stream
        x32, dy32 = int(x.dtype == torch.float32), int(dy.dtype == torch.float32)
        p = lambda t: None if t is None else t.data_ptr()
        if self.split:                                     # fp16-side tensors are plane pairs: pass each one's plane pitch
            lo = lambda t, is32=0: 0 if (t is None or is32) else t.stride(0)
            _lib.check(_lib.lib().advh_layernorm_bwd_split(
                x.data_ptr(), x32, lo(x, x32), dy.data_ptr(), dy32, lo(dy, dy32), ln.g.data_ptr(), ln.b.data_ptr(), int(gelu), p(add),
                p(dact), lo(dact), p(out_f), p(out_h), lo(out_h), M, ln.C, eps, remap[0], remap[1], st), "advh_layernorm_bwd_split")
            return
        _lib.check(_lib.lib().advh_layernorm_bwd(
            x.data_ptr(), x32, dy.data_ptr(), dy32, ln.g.data_ptr(), ln.b.data_ptr(), int(gelu), p(add), p(dact), p(out_f), p(out_h), M,
            ln.C, eps, remap[0], remap[1], st), "advh_layernorm_bwd")

    def _att_bwd(self, qkv, dctx, dqkv, B, T, H, heads, st):
        lib = _lib.lib()
        if self.split:
            _lib.check(lib.advh_attention_bwd_split(qkv.data_ptr(), qkv.stride(0), dctx.data_ptr(), dctx.stride(0), dqkv.data_ptr(),
                                                    dqkv.stride(0), B, T, H, heads, st), "advh_attention_bwd_split")
        else:
            _lib.check(lib.advh_attention_bwd_f16(qkv.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(), B, T, H, heads, st),
                       "advh_attention_bwd_f16")

    def backward(self, loss_scale: float = 4096.0, seed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """d logit / d wave for the clips of the last ``forward`` call: ``[B, n_in]`` fp32.  With ``seed [B]``
        (dL/d logit per clip) the result is dL/d wave instead (vector-Jacobian product: LMACLoss backward)."""
        emb, cfg, lib, sp = self.emb, self.cfg, _lib.lib(), self.split
        wave, B, n_in, L = self._last
        w = self._workspace(B, L)
        f = w["f"]
        st = torch.cuda.current_stream().cuda_stream
        Ls, P, T, M, H = f["Ls"], f["P"], f["T"], f["M"], cfg.hidden_size
        C, nfe, nl = cfg.conv_dim, len(f["Ls"]), emb.nl
        da, db, d16, t16 = w["da"], w["db"], w["d16"], w["t16"]
        heads = cfg.num_attention_heads
        if seed is None:
            w["dlogit"].fill_(loss_scale)
        else:
            w["dlogit"].copy_(seed.reshape(-1).to(w["dlogit"].dtype) * loss_scale)
        if sp:
            _lib.check(lib.advh_pool_logreg_bwd_split(emb.coef.data_ptr(), w["dlogit"].data_ptr(), da.data_ptr(), d16.data_ptr(),
                                                      d16.stride(0), B, T, H, st), "advh_pool_logreg_bwd_split")
        else:
            _lib.check(lib.advh_pool_logreg_bwd(emb.coef.data_ptr(), w["dlogit"].data_ptr(), da.data_ptr(), d16.data_ptr(), B, T, H, st),
                       "advh_pool_logreg_bwd")
        if self._final_ln:
            self._ln_bwd(emb.enc_ln, w["x"][nl], da, M, out_f=db, out_h=d16)
            da, db = db, da
        for l in range(nl - 1, -1, -1):
            bl = w["layers"][l]
            if self.stable:                                # da = d x_{l+1} (fp32), d16 its fp16 copy
                bl["ff2"].run(d16, out_h=w["dI"], dact_src=w["g1"][l])
                bl["ff1"].run(w["dI"], out_h=t16)                                             # d LN2(m)
                self._ln_bwd(emb.ln2[l], w["m"][l], t16, M, out_f=db, out_h=d16, add=da)      # db = d m
                bl["out"].run(d16, out_h=w["dctx"])
                self._att_bwd(w["qkv"][l], w["dctx"], w["dqkv"], B, T, H, heads, st)
                bl["qkv"].run(w["dqkv"], out_h=t16)                                           # d LN1(x_l)
                self._ln_bwd(emb.ln1[l], w["x"][l], t16, M, out_f=da, out_h=d16, add=db)      # da = d x_l
            else:
                self._ln_bwd(emb.ln2[l], w["s2"][l], da, M, out_f=db, out_h=d16)              # db = d s2
                bl["ff2"].run(d16, out_h=w["dI"], dact_src=w["g1"][l])                        # d(pre-GELU)
                bl["ff1"].run(w["dI"], out_f=da, resid=db)                                    # da = d m
                self._ln_bwd(emb.ln1[l], w["s1"][l], da, M, out_f=db, out_h=d16)              # db = d s1
                bl["out"].run(d16, out_h=w["dctx"])
                self._att_bwd(w["qkv"][l], w["dctx"], w["dqkv"], B, T, H, heads, st)
                bl["qkv"].run(w["dqkv"], out_f=da, resid=db)                                  # da = d x_l
        if not self.stable:
            self._ln_bwd(emb.enc_ln, w["h1"], da, M, out_f=db)                                # d h1
            da, db = db, da
        K, Gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        if sp:
            _lib.check(lib.advh_posconv_gather_bwd_split(da.data_ptr(), f["xg"].data_ptr(), f["xg"].stride(0), B, T, H, Gp, K, K // 2 - 1,
                                                         w["pc"].data_ptr(), w["pc"].stride(0), st), "advh_posconv_gather_bwd_split")
        else:
            _lib.check(lib.advh_posconv_gather(da.data_ptr(), f["xg"].data_ptr(), B, T, H, Gp, K, K // 2 - 1, w["pc"].data_ptr(), st),
                       "advh_posconv_gather")
        w["pos"].run(f["xg"], out_f=db, out_h=d16, resid=da)                                  # d h0
        w["proj"].run(d16, out_h=w["dfeatn"])
        dz = w["dz"]
        last = nfe - 1
        body = lambda i: dz[i][..., C[i]:]                                                    # skip the guard row
        zb = lambda i: w["z"][i] if i == last else self._body(w["z"][i], i)
        if self.layer_mode:
            self._ln_bwd(emb.fp_ln, w["feat"], w["dfeatn"], M, out_h=w["dfeat"])
            self._ln_bwd(emb.fe_ln[last], zb(last), w["dfeat"], M, out_h=body(last), remap=(T, P[last]), gelu=True, eps=1e-5)
        else:
            self._ln_bwd(emb.fp_ln, w["feat"], w["dfeatn"], M, out_h=body(last), dact=zb(last), remap=(T, P[last]))
        for i in range(last, 0, -1):
            plan, nt = w["fe"][i - 1]
            a0 = dz[i] if nt == 2 else body(i)
            if self.layer_mode:
                dyb = self._body(w["dyb"], 0)                                                 # level-0 sized scratch: large enough for every level
                plan.run(a0, out_h=dyb)
                self._ln_bwd(emb.fe_ln[i - 1], zb(i - 1), dyb, B * P[i - 1], out_h=body(i - 1), gelu=True, eps=1e-5)
            else:
                plan.run(a0, out_h=body(i - 1), dact_src=zb(i - 1) if i > 1 else None)
        if not self.layer_mode:
            ln0 = emb.fe_ln[0]
            if sp:
                _lib.check(lib.advh_w2v2_frontend_bwd_group_split(
                    wave.data_ptr(), wave.stride(0), n_in, B, L, emb.w0.data_ptr(), ln0.g.data_ptr(), f["stats"].data_ptr(),
                    f["norm"].data_ptr(), f["mr"].data_ptr(), body(0).data_ptr(), dz[0].stride(0), w["part"].data_ptr(),
                    w["sums"].data_ptr(), body(0).data_ptr(), dz[0].stride(0), Ls[0], P[0], C[0], st), "advh_w2v2_frontend_bwd_group_split")
            else:
                _lib.check(lib.advh_w2v2_frontend_bwd_group(
                    wave.data_ptr(), wave.stride(0), n_in, B, L, emb.w0.data_ptr(), ln0.g.data_ptr(), f["stats"].data_ptr(),
                    f["norm"].data_ptr(), f["mr"].data_ptr(), body(0).data_ptr(), w["part"].data_ptr(), w["sums"].data_ptr(),
                    body(0).data_ptr(), Ls[0], P[0], C[0], st), "advh_w2v2_frontend_bwd_group")
        w["g_plan"].run(body(0), out_f=w["g"])
        dx = torch.empty((B, n_in), dtype=torch.float32, device=wave.device)
        _lib.check(lib.advh_wave_bwd(w["g"].data_ptr(), wave.data_ptr(), wave.stride(0), n_in, B, L, f["stats"].data_ptr(),
                                     w["dxh"].data_ptr(), w["wpart"].data_ptr(), 1, 1.0 / loss_scale, dx.data_ptr(), n_in,
                                     Ls[0], P[0], st), "advh_wave_bwd")
        return dx
