"""Oracle: frozen wav2vec2 embedder + logreg head.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates HF ``Wav2Vec2Model.forward`` in eval mode (transformers 5.15.0,
models/wav2vec2/modeling_wav2vec2.py) as plain functional torch over an HF-named state dict,
so it runs on the GPU box without ``transformers`` and is differentiable w.r.t. the waveform
(needed by the attribution oracle).  ``cfg`` is a ``synthetic.EmbedderConfig``.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from .signal_ref import zero_mean_unit_var_norm


def pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """weight_norm(dim=2) folded: w = g * v / ||v||_{dims 0,1}  (modeling_wav2vec2.py:326-357)."""
    g = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def feature_encoder(x: torch.Tensor, sd, cfg) -> torch.Tensor:
    """modeling_wav2vec2.py:254-323, 382-419.  ``[B, L] -> [B, C, T]``."""
    h = x[:, None]
    for i, (k, s) in enumerate(zip(cfg.conv_kernel, cfg.conv_stride)):
        p = f"feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd.get(p + "conv.bias"), stride=s)
        if cfg.feat_extract_norm == "group" and i == 0:
            C = h.shape[1]
            h = F.group_norm(h, C, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps=1e-5)
        elif cfg.feat_extract_norm == "layer":
            h = F.layer_norm(h.transpose(-2, -1), (h.shape[1],), sd[p + "layer_norm.weight"],
                             sd[p + "layer_norm.bias"], eps=1e-5).transpose(-2, -1)
        h = F.gelu(h)
    return h


def attention(h: torch.Tensor, sd, p: str, nheads: int) -> torch.Tensor:
    """modeling_wav2vec2.py:438-548 -- softmax(Q K^T / sqrt(d)) V, no mask (math path,
    train_addvisor.py:21-23)."""
    B, T, H = h.shape
    d = H // nheads
    q = F.linear(h, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, nheads, d).transpose(1, 2)
    k = F.linear(h, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, nheads, d).transpose(1, 2)
    v = F.linear(h, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, nheads, d).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * d ** -0.5, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, H)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def feed_forward(h, sd, p):
    """modeling_wav2vec2.py:551-572."""
    h = F.gelu(F.linear(h, sd[p + "intermediate_dense.weight"], sd[p + "intermediate_dense.bias"]))
    return F.linear(h, sd[p + "output_dense.weight"], sd[p + "output_dense.bias"])


def hidden_states(wave_normed: torch.Tensor, sd, cfg, upto: int | None = None) -> List[torch.Tensor]:
    """``Wav2Vec2Model(input_values, output_hidden_states=True).hidden_states[: upto+1]``
    (modeling_wav2vec2.py:1319-1375, 657-802).  Only the first ``upto`` layers run."""
    eps = cfg.layer_norm_eps
    nl = cfg.num_hidden_layers if upto is None else min(upto, cfg.num_hidden_layers)
    feats = feature_encoder(wave_normed, sd, cfg).transpose(1, 2)                    # [B,T,C]
    C, H = feats.shape[-1], cfg.hidden_size
    h = F.layer_norm(feats, (C,), sd["feature_projection.layer_norm.weight"],
                     sd["feature_projection.layer_norm.bias"], eps)
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    pos = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"],
                   padding=K // 2, groups=G)
    if K % 2 == 0:
        pos = pos[:, :, :-1]                                                         # SamePad :359-379
    h = h + F.gelu(pos).transpose(1, 2)
    out: List[torch.Tensor] = []
    if not cfg.do_stable_layer_norm:                                                 # post-LN :657-726
        h = F.layer_norm(h, (H,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
        for l in range(nl):
            out.append(h)
            p = f"encoder.layers.{l}."
            h = h + attention(h, sd, p + "attention.", cfg.num_attention_heads)
            h = F.layer_norm(h, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
            h = h + feed_forward(h, sd, p + "feed_forward.")
            h = F.layer_norm(h, (H,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
        out.append(h)
    else:                                                                            # pre-LN :729-802
        for l in range(nl):
            out.append(h)
            p = f"encoder.layers.{l}."
            a = F.layer_norm(h, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
            h = h + attention(a, sd, p + "attention.", cfg.num_attention_heads)
            f = F.layer_norm(h, (H,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
            h = h + feed_forward(f, sd, p + "feed_forward.")
        if nl == cfg.num_hidden_layers:
            h = F.layer_norm(h, (H,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
        out.append(h)
    return out


def extract_features(waveforms: torch.Tensor, sd, cfg) -> torch.Tensor:
    """audioprocessor.py:69-77 -- normalise -> wav2vec2 -> hidden_states[layer_index].squeeze(0)
    (``[T,H]`` when B == 1)."""
    x = zero_mean_unit_var_norm(waveforms)
    return hidden_states(x, sd, cfg, upto=cfg.layer_index)[cfg.layer_index].squeeze(0)


def logreg(x: torch.Tensor, coef, intercept):
    """classifier_embedder.py:21-38 -- Linear(H,1) from sklearn coef_/intercept_; (logits, sigmoid)."""
    w = torch.as_tensor(coef, dtype=torch.float32)
    b = torch.as_tensor(intercept, dtype=torch.float32)
    logits = F.linear(x, w, b)
    return logits, torch.sigmoid(logits)


def classify(waveforms: torch.Tensor, sd, cfg, coef, intercept):
    """Pooling at the call sites LMAC_metrics.py:130,146,156 / train_addvisor.py:254-255:
    mean over the time axis per example (the intended semantics of captum_saliency.py:90-100,
    SURVEY D6), then the logreg.  ``[B, L] -> (logits [B,1], probs [B,1])``."""
    x = zero_mean_unit_var_norm(waveforms)
    h = hidden_states(x, sd, cfg, upto=cfg.layer_index)[cfg.layer_index]
    return logreg(h.mean(dim=1), coef, intercept)
