"""GPU parity of the wav2vec2 embedder + logreg head (HIP kernels) against the CPU oracle.

Stated tolerance (fp16 GEMM operands, fp32 accumulation and fp32 residual stream / norms):
  hidden_states[9]: max |err| <= 3e-2 (values are O(1) after LayerNorm), mean |err| <= 3e-3;
  classifier logits: |err| <= 1e-2.
"""
import numpy as np
import pytest
import torch

from addvisor_hip import synthetic as syn
from addvisor_hip.embedder import HipEmbedder
from oracle import wav2vec2_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL_HID_MAX, TOL_HID_MEAN, TOL_LOGIT = 3e-2, 3e-3, 1e-2


def run_case(cfg, waves, dev, length=None):
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    emb = HipEmbedder(cfg, sd, coef, icpt, dev, precision="f16")
    hid, logit, prob = emb.forward(waves.to(dev), length)
    w = waves if length is None else torch.nn.functional.pad(waves, (0, max(0, length - waves.shape[1])))[:, :length]
    x = wav2vec2_ref.zero_mean_unit_var_norm(w)
    ref_h = wav2vec2_ref.hidden_states(x, sd, cfg, upto=cfg.layer_index)[min(cfg.layer_index, cfg.num_hidden_layers)]
    ref_logit, ref_prob = wav2vec2_ref.logreg(ref_h.mean(1), coef, icpt)
    err = (hid.cpu() - ref_h).abs()
    print(f"hidden max err {err.max():.3e} mean {err.mean():.3e} | ref absmax {ref_h.abs().max():.2f} | "
          f"logit err {(logit.cpu() - ref_logit).abs().max():.3e}")
    assert err.max().item() <= TOL_HID_MAX and err.mean().item() <= TOL_HID_MEAN
    assert (logit.cpu() - ref_logit).abs().max().item() <= TOL_LOGIT
    assert (prob.cpu() - ref_prob).abs().max().item() <= TOL_LOGIT
    return hid, logit


@pytest.mark.parametrize("stable", [False, True])
def test_tiny_embedder(gpu_device, stable, golden):
    cfg = syn.tiny_config(stable)
    w = syn.make_clips(2, 16000, seed=31)
    hid, _ = run_case(cfg, w, gpu_device)
    g = golden(f"embedder_tiny_{'layer' if stable else 'group'}.npz")     # the reference's own extract_features
    assert (hid.cpu() - torch.from_numpy(g["feats_b2"])).abs().max().item() <= TOL_HID_MAX


def test_tiny_depth9_final_layernorm(gpu_device, golden):
    cfg = syn.tiny_config(True, num_hidden_layers=9)
    hid, _ = run_case(cfg, syn.make_clips(2, 16000, seed=31), gpu_device)
    g = golden("embedder_tiny_layer_depth9.npz")
    assert (hid.cpu() - torch.from_numpy(g["feats_b2"])).abs().max().item() <= TOL_HID_MAX


def test_tiny_pad_and_crop(gpu_device):
    cfg = syn.tiny_config(False)
    run_case(cfg, syn.make_clips(3, 9000, seed=5), gpu_device, length=12000)   # zero-padded tail
    run_case(cfg, syn.make_clips(3, 20000, seed=6), gpu_device, length=16000)  # cropped


def test_base_embedder_4s(gpu_device, golden):
    cfg = syn.base_config()
    w = syn.make_clips(2, 64000)
    hid, _ = run_case(cfg, w, gpu_device)
    g = golden("embedder_base_4s.npz")
    assert tuple(hid.shape[1:]) == tuple(g["shape"])
    assert (hid[0, :8, :16].cpu() - torch.from_numpy(g["corner"])).abs().max().item() <= TOL_HID_MAX
    assert (hid[0].mean(0).cpu() - torch.from_numpy(g["pooled"])).abs().max().item() <= 5e-3


def test_batch_invariance(gpu_device):
    """Clips are independent: a clip's result must not depend on its batch neighbours (bit-exact)."""
    cfg = syn.tiny_config(False)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    emb = HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision="f16")
    w = syn.make_clips(5, 16000, seed=8).to(gpu_device)
    h5, l5, _ = emb.forward(w)
    h2, l2, _ = emb.forward(w[1:3].contiguous())
    assert torch.equal(h5[1:3], h2) and torch.equal(l5[1:3], l2)


def test_xlsr_shaped_embedder(gpu_device):
    """The reference's own embedder family (XLS-R-2B: hidden 1920, 16 heads -> head_dim 120, layer-norm feature
    extractor, pre-LN encoder) at reduced width/depth: hidden 240 / 2 heads keeps head_dim = 120."""
    cfg = syn.tiny_config(True, hidden_size=240, num_attention_heads=2, intermediate_size=480,
                          num_conv_pos_embedding_groups=2, num_hidden_layers=10)
    run_case(cfg, syn.make_clips(2, 16000, seed=33), gpu_device)
