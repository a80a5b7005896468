"""Deterministic synthetic clips and weights.

No checkpoint of the reference exists offline (the XLS-R embedder, the logreg
``.joblib``, the U-Net ``.pth`` and the SpeechBrain vocoder are all private or
remote: classifier_embedder.py:12-16, LMAC_metrics.py:21, hifigan.py:106-110),
so every parity and bench run uses weights and clips generated here from a
seed.  numpy's PCG64 stream is stable across platforms and numpy versions, so
the same seed gives the same tensors in this container and on the GPU box.

All tensors come back as torch fp32 CPU tensors keyed by the *reference's*
parameter names (HF ``Wav2Vec2Model`` names for the embedder, addvisor.py:31-60
names for the U-Net), so they load into the reference modules with
``load_state_dict`` as well as into the HIP runtime.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Tuple

import numpy as np
import torch

CLIP_SEED = 1234
WEIGHT_SEED = 4321


# --------------------------------------------------------------------------- clips
def make_clips(n: int, length: int = 64000, sr: int = 16000, seed: int = CLIP_SEED,
               first: int = 0) -> torch.Tensor:
    """``n`` clips ``[n, length]`` fp32 in [-1, 1]: 5 sinusoids in 80..7600 Hz + 0.05 N(0,1).

    Clip ``i`` depends only on ``(seed, first + i)`` so shards of a data set can be
    generated independently on every rank.
    """
    out = np.empty((n, length), dtype=np.float32)
    t = np.arange(length, dtype=np.float64) / sr
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64([seed, first + i]))
        f = rng.uniform(80.0, 7600.0, size=5)
        a = rng.uniform(0.05, 0.3, size=5)
        ph = rng.uniform(0.0, 2 * math.pi, size=5)
        x = (a[:, None] * np.sin(2 * math.pi * f[:, None] * t[None, :] + ph[:, None])).sum(0)
        x = x + 0.05 * rng.standard_normal(length)
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- embedder config
@dataclasses.dataclass(frozen=True)
class EmbedderConfig:
    """The subset of HF ``Wav2Vec2Config`` the frozen embedder needs
    (transformers/models/wav2vec2/configuration_wav2vec2.py:163-219)."""

    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"          # "group" (base) | "layer" (large / xls-r)
    do_stable_layer_norm: bool = False        # False: post-LN (base); True: pre-LN
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    layer_index: int = 9                      # audioprocessor.py:77  hidden_states[9]

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def frames(self, n_samples: int) -> int:
        n = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
        return n

    def hf_kwargs(self) -> dict:
        """kwargs for ``transformers.Wav2Vec2Config`` (tests / fixture generation only)."""
        return dict(
            hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
            num_attention_heads=self.num_attention_heads, intermediate_size=self.intermediate_size,
            conv_dim=list(self.conv_dim), conv_kernel=list(self.conv_kernel),
            conv_stride=list(self.conv_stride), conv_bias=self.conv_bias,
            feat_extract_norm=self.feat_extract_norm, do_stable_layer_norm=self.do_stable_layer_norm,
            num_conv_pos_embeddings=self.num_conv_pos_embeddings,
            num_conv_pos_embedding_groups=self.num_conv_pos_embedding_groups,
            layer_norm_eps=self.layer_norm_eps, num_feat_extract_layers=len(self.conv_dim),
            hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0,
            layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.0, vocab_size=32,
            hidden_act="gelu", feat_extract_activation="gelu",
        )


def base_config(**kw) -> EmbedderConfig:
    """wav2vec2-base: BASELINE config 2."""
    return EmbedderConfig(**kw)


def large_config(**kw) -> EmbedderConfig:
    """wav2vec2-large (layer-norm FE, pre-LN encoder): BASELINE config 5."""
    d = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True)
    d.update(kw)
    return EmbedderConfig(**d)


def xlsr2b_config(**kw) -> EmbedderConfig:
    """XLS-R-2B shape (hidden 1920): the reference's own embedder (classifier_embedder.py:13-16, 25)."""
    d = dict(hidden_size=1920, num_hidden_layers=48, num_attention_heads=16, intermediate_size=7680,
             conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True)
    d.update(kw)
    return EmbedderConfig(**d)


def tiny_config(stable: bool = False, **kw) -> EmbedderConfig:
    """A seconds-on-CPU shape used by the golden fixtures: 10 layers so ``hidden_states[9]`` exists."""
    d = dict(hidden_size=64, num_hidden_layers=10, num_attention_heads=2, intermediate_size=128,
             conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2,
             conv_bias=stable, feat_extract_norm="layer" if stable else "group",
             do_stable_layer_norm=stable)
    d.update(kw)
    return EmbedderConfig(**d)


# --------------------------------------------------------------------------- weights
class _Gen:
    def __init__(self, seed: int, tag: int):
        self.rng = np.random.Generator(np.random.PCG64([seed, tag]))

    def uniform(self, shape, bound: float) -> torch.Tensor:
        return torch.from_numpy(self.rng.uniform(-bound, bound, size=shape).astype(np.float32))

    def normal(self, shape, std: float) -> torch.Tensor:
        return torch.from_numpy((std * self.rng.standard_normal(size=shape)).astype(np.float32))

    def affine(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """LayerNorm / GroupNorm / BatchNorm affine: gamma in [0.8, 1.2], beta in [-0.1, 0.1]."""
        return (torch.from_numpy(self.rng.uniform(0.8, 1.2, size=n).astype(np.float32)),
                torch.from_numpy(self.rng.uniform(-0.1, 0.1, size=n).astype(np.float32)))


def embedder_weights(cfg: EmbedderConfig, seed: int = WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """State dict with HF ``Wav2Vec2Model`` names (transformers/.../modeling_wav2vec2.py:254-802).

    Only layers ``0 .. cfg.num_hidden_layers-1`` are generated; the HIP runtime and the oracle
    run the first ``layer_index`` of them.
    """
    g = _Gen(seed, 1)
    sd: Dict[str, torch.Tensor] = {}
    cin = 1
    for i, (co, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"] = g.uniform((co, cin, k), math.sqrt(6.0 / (cin * k)))
        if cfg.conv_bias:
            sd[p + "conv.bias"] = g.uniform((co,), 0.05)
        if (cfg.feat_extract_norm == "group" and i == 0) or cfg.feat_extract_norm == "layer":
            sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"] = g.affine(co)
        cin = co
    H, C = cfg.hidden_size, cfg.conv_dim[-1]
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = g.affine(C)
    sd["feature_projection.projection.weight"] = g.uniform((H, C), math.sqrt(3.0 / C))
    sd["feature_projection.projection.bias"] = g.uniform((H,), 0.05)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    v = g.uniform((H, H // G, K), math.sqrt(3.0 / (K * H // G)))
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = v
    # weight_norm(dim=2): g has shape (1, 1, K); start from the norm of v, perturbed
    gn = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = gn * (
        1.0 + g.uniform((1, 1, K), 0.2))
    sd["encoder.pos_conv_embed.conv.bias"] = g.uniform((H,), 0.05)
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = g.affine(H)
    I = cfg.intermediate_size
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{nm}.weight"] = g.uniform((H, H), math.sqrt(3.0 / H))
            sd[p + f"attention.{nm}.bias"] = g.uniform((H,), 0.05)
        sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"] = g.affine(H)
        sd[p + "feed_forward.intermediate_dense.weight"] = g.uniform((I, H), math.sqrt(3.0 / H))
        sd[p + "feed_forward.intermediate_dense.bias"] = g.uniform((I,), 0.05)
        sd[p + "feed_forward.output_dense.weight"] = g.uniform((H, I), math.sqrt(3.0 / I))
        sd[p + "feed_forward.output_dense.bias"] = g.uniform((H,), 0.05)
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = g.affine(H)
    return sd


def logreg_weights(hidden: int, seed: int = WEIGHT_SEED, scale: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """sklearn-style ``coef_ (1,H)``, ``intercept_ (1,)`` (classifier_embedder.py:25-33).

    ``coef ~ N(0, 4/H)``: large enough that |logit| is O(1) on random-weight features, so the
    ``p > 0.5`` labels of the fidelity metric are not all inside the rounding band.
    """
    rng = np.random.Generator(np.random.PCG64([seed, 2]))
    coef = (scale * math.sqrt(4.0 / hidden) * rng.standard_normal((1, hidden))).astype(np.float64)
    intercept = np.array([0.1 * rng.standard_normal()], dtype=np.float64)
    return coef, intercept


# U-Net layer table: (state_dict prefix, kind, cin, cout, kernel(h,w))  -- addvisor.py:31-60
def unet_layer_table() -> List[Tuple[str, str, int, int, Tuple[int, int]]]:
    t: List[Tuple[str, str, int, int, Tuple[int, int]]] = []

    def block(name, cin, cout, k):
        t.append((f"{name}.block.0", "conv", cin, cout, k))
        t.append((f"{name}.block.1", "bn", cout, cout, (0, 0)))
        t.append((f"{name}.block.3", "conv", cout, cout, (3, 3)))
        t.append((f"{name}.block.4", "bn", cout, cout, (0, 0)))

    block("e1", 1, 32, (5, 3))
    block("e2", 32, 64, (5, 3))
    block("e3", 64, 128, (3, 3))
    block("e4", 128, 256, (3, 3))
    t.append(("bottleneck.0", "conv", 256, 512, (3, 3)))
    t.append(("bottleneck.1", "bn", 512, 512, (0, 0)))
    t.append(("bottleneck.3", "conv", 512, 512, (3, 3)))
    t.append(("bottleneck.4", "bn", 512, 512, (0, 0)))
    t.append(("up4", "convT", 512, 256, (2, 2)))
    block("d4", 384, 256, (3, 3))
    t.append(("up3", "convT", 256, 128, (2, 2)))
    block("d3", 192, 128, (3, 3))
    t.append(("up2", "convT", 128, 64, (2, 1)))
    block("d2", 96, 64, (3, 3))
    t.append(("up1", "convT", 64, 32, (2, 1)))
    block("d1", 33, 32, (3, 3))
    t.append(("mask_head.0", "conv", 32, 1, (1, 1)))
    return t


def unet_weights(seed: int = WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """State dict with the reference's U-Net parameter names (addvisor.py:31-60)."""
    g = _Gen(seed, 3)
    sd: Dict[str, torch.Tensor] = {}
    for name, kind, cin, cout, k in unet_layer_table():
        if kind == "conv":
            fan = cin * k[0] * k[1]
            sd[name + ".weight"] = g.uniform((cout, cin, k[0], k[1]), math.sqrt(6.0 / fan) * 0.9)
            sd[name + ".bias"] = g.uniform((cout,), 0.05)
        elif kind == "convT":
            sd[name + ".weight"] = g.uniform((cin, cout, k[0], k[1]), math.sqrt(3.0 / cin))
            sd[name + ".bias"] = g.uniform((cout,), 0.05)
        else:  # bn
            sd[name + ".weight"], sd[name + ".bias"] = g.affine(cout)
            sd[name + ".running_mean"] = g.uniform((cout,), 0.1)
            sd[name + ".running_var"] = torch.from_numpy(
                g.rng.uniform(0.5, 1.5, size=cout).astype(np.float32))
            sd[name + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.long)
    return sd


# --------------------------------------------------------------------------- HiFi-GAN V1
@dataclasses.dataclass(frozen=True)
class HifiganConfig:
    """HiFi-GAN V1 generator hyper-parameters (Kong et al. 2020, config_v1) at 16 kHz / hop 256
    (hifigan.py:106-110 loads SpeechBrain's ``tts-hifigan-libritts-16kHz``; its source is absent)."""

    in_channels: int = 80
    upsample_initial_channel: int = 512
    upsample_rates: Tuple[int, ...] = (8, 8, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4)
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilations: Tuple[int, ...] = (1, 3, 5)
    leaky_slope: float = 0.1
    pre_kernel: int = 7
    post_kernel: int = 7

    @property
    def hop(self) -> int:
        h = 1
        for r in self.upsample_rates:
            h *= r
        return h


def hifigan_tiny_config() -> HifiganConfig:
    return HifiganConfig(in_channels=16, upsample_initial_channel=128)      # stages 64 / 32 / 16 / 8 channels


def hifigan_weights(cfg: HifiganConfig = HifiganConfig(), seed: int = WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """Weight-norm already folded (inference form). Names:
    ``conv_pre``, ``ups.{i}``, ``resblocks.{i*3+j}.convs1.{d}`` / ``convs2.{d}``, ``conv_post``."""
    g = _Gen(seed, 4)
    sd: Dict[str, torch.Tensor] = {}
    ch = cfg.upsample_initial_channel
    sd["conv_pre.weight"] = g.uniform((ch, cfg.in_channels, cfg.pre_kernel),
                                      math.sqrt(3.0 / (cfg.in_channels * cfg.pre_kernel)))
    sd["conv_pre.bias"] = g.uniform((ch,), 0.05)
    for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        co = ch // 2
        # ConvTranspose1d weight [Cin, Cout, K]; every output sees K/r taps x Cin inputs
        sd[f"ups.{i}.weight"] = g.uniform((ch, co, k), math.sqrt(6.0 / (ch * k / r)))
        sd[f"ups.{i}.bias"] = g.uniform((co,), 0.05)
        for j, rk in enumerate(cfg.resblock_kernel_sizes):
            for d in range(len(cfg.resblock_dilations)):
                p = f"resblocks.{i * len(cfg.resblock_kernel_sizes) + j}."
                sd[p + f"convs1.{d}.weight"] = g.uniform((co, co, rk), math.sqrt(3.0 / (co * rk)))
                sd[p + f"convs1.{d}.bias"] = g.uniform((co,), 0.05)
                sd[p + f"convs2.{d}.weight"] = g.uniform((co, co, rk), 0.5 * math.sqrt(3.0 / (co * rk)))
                sd[p + f"convs2.{d}.bias"] = g.uniform((co,), 0.05)
        ch = co
    sd["conv_post.weight"] = g.uniform((1, ch, cfg.post_kernel), 0.1 * math.sqrt(3.0 / (ch * cfg.post_kernel)))   # keeps tanh out of saturation
    sd["conv_post.bias"] = g.uniform((1,), 0.05)
    return sd
