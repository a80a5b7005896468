"""Process-wide model state behind the drop-in modules (classifier_embedder / audioprocessor / ...).

The reference loads a private XLS-R checkpoint, a logreg ``.joblib`` and a U-Net ``.pth`` at import time
(classifier_embedder.py:12-16, LMAC_metrics.py:21).  Here nothing is fetched or loaded at import:
models are built lazily on first use, from local files named by environment variables or -- the
default, since none of those artefacts exist offline -- from the seeded synthetic generator.

    ADDVISOR_EMBEDDER     base | large | tiny | tiny_layer (synthetic weights)   or a local HF directory
                          (config.json + model.safetensors / pytorch_model.bin of a Wav2Vec2Model)
    ADDVISOR_LOGREG       .npz with coef_ / intercept_, or a sklearn .joblib       (default: synthetic)
    ADDVISOR_LAYER_INDEX  hidden-state index returned by extract_features            (default 9)
    ADDVISOR_PRECISION    f32 (default: fp32-class split-format kernels, the reference's arithmetic class) | f16
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import synthetic as syn

_state = {}


def device() -> torch.device:
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _load_hf_dir(path: str):
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    cfg = syn.EmbedderConfig(
        hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
        num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
        conv_dim=tuple(c["conv_dim"]), conv_kernel=tuple(c["conv_kernel"]), conv_stride=tuple(c["conv_stride"]),
        conv_bias=c.get("conv_bias", False), feat_extract_norm=c.get("feat_extract_norm", "group"),
        do_stable_layer_norm=c.get("do_stable_layer_norm", False),
        num_conv_pos_embeddings=c.get("num_conv_pos_embeddings", 128),
        num_conv_pos_embedding_groups=c.get("num_conv_pos_embedding_groups", 16),
        layer_norm_eps=c.get("layer_norm_eps", 1e-5), layer_index=int(os.environ.get("ADDVISOR_LAYER_INDEX", 9)))
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    sd = {k.replace("wav2vec2.", "", 1) if k.startswith("wav2vec2.") else k: v for k, v in sd.items()}
    # older checkpoints store weight_norm as weight_g / weight_v
    for old, new in (("weight_g", "parametrizations.weight.original0"), ("weight_v", "parametrizations.weight.original1")):
        k = "encoder.pos_conv_embed.conv." + old
        if k in sd:
            sd["encoder.pos_conv_embed.conv." + new] = sd.pop(k)
    return cfg, sd


def embedder_config_and_weights():
    if "emb" not in _state:
        spec = os.environ.get("ADDVISOR_EMBEDDER", "base")
        li = int(os.environ.get("ADDVISOR_LAYER_INDEX", 9))
        named = {"base": lambda: syn.base_config(layer_index=li), "large": lambda: syn.large_config(layer_index=li),
                 "tiny": lambda: syn.tiny_config(False, layer_index=li), "tiny_layer": lambda: syn.tiny_config(True, layer_index=li)}
        if spec in named:
            cfg = named[spec]()
            _state["emb"] = (cfg, syn.embedder_weights(cfg))
        elif os.path.isdir(spec):
            _state["emb"] = _load_hf_dir(spec)
        else:
            raise FileNotFoundError(f"ADDVISOR_EMBEDDER={spec!r} is neither a synthetic preset nor a local directory")
    return _state["emb"]


class SkLogReg:
    """What ``joblib.load`` returns in the reference: an object with ``coef_ (1,H)`` and ``intercept_ (1,)``."""

    def __init__(self, coef_, intercept_):
        self.coef_, self.intercept_ = np.asarray(coef_, dtype=np.float64), np.asarray(intercept_, dtype=np.float64)


def classifier() -> SkLogReg:
    if "clf" not in _state:
        path = os.environ.get("ADDVISOR_LOGREG")
        cfg, _ = embedder_config_and_weights()
        if not path:
            _state["clf"] = SkLogReg(*syn.logreg_weights(cfg.hidden_size))
        elif path.endswith(".npz"):
            z = np.load(path)
            _state["clf"] = SkLogReg(z["coef_"], z["intercept_"])
        else:
            import joblib
            m = joblib.load(path)
            _state["clf"] = SkLogReg(m.coef_, m.intercept_)
    return _state["clf"]


def hip_embedder():
    """The HIP embedder singleton (needs a GPU; raises otherwise -- there is no CPU path)."""
    if "hip_emb" not in _state:
        if not torch.cuda.is_available():
            raise RuntimeError("the ADDvisor HIP path needs an AMD GPU (no CPU fallback exists)")
        from .embedder import HipEmbedder
        cfg, sd = embedder_config_and_weights()
        clf = classifier()
        _state["hip_emb"] = HipEmbedder(cfg, sd, clf.coef_, clf.intercept_, device())
    return _state["hip_emb"]


def hip_embedder_grad():
    """Forward-with-saves + input-gradient engine on the HIP embedder singleton (LMACLoss backward, attributions)."""
    if "hip_eg" not in _state:
        from .embedder_grad import EmbedderGrad
        _state["hip_eg"] = EmbedderGrad(hip_embedder())
    return _state["hip_eg"]


def reset():
    _state.clear()
