"""Oracle: Captum-style gradient attributions of the waveform -> logit model.
TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: ``captum`` is absent, so this
follows the call sites captum_saliency.py:116-118,131-143 and Captum's published algorithms
(Sundararajan et al. 2017 for IG; Captum defaults n_steps=50, method="gausslegendre",
zero baseline, multiply_by_inputs=True, scaled inputs concatenated step-major)."""
from __future__ import annotations

import numpy as np
import torch

from . import wav2vec2_ref as W


def model_logit(waves, emb_sd, emb_cfg, coef, intercept):
    """captum_saliency.py:84-100 with per-example time pooling.  ``[B,L] -> [B,1]``."""
    return W.classify(waves, emb_sd, emb_cfg, coef, intercept)[0]


def input_gradient(waves, emb_sd, emb_cfg, coef, intercept):
    with torch.enable_grad():
        x = waves.clone().detach().requires_grad_(True)
        out = model_logit(x, emb_sd, emb_cfg, coef, intercept)
        (g,) = torch.autograd.grad(out.sum(), x)
    return g


def saliency(waves, *model):
    """captum.attr.Saliency(abs=True): |dF/dx|."""
    return input_gradient(waves, *model).abs()


def input_x_gradient(waves, *model):
    """captum.attr.InputXGradient (the active choice, captum_saliency.py:117): x * dF/dx."""
    return waves * input_gradient(waves, *model)


def gauss_legendre(n_steps: int):
    """Captum ``approximation_method="gausslegendre"``: alphas = (1+x)/2, step sizes = w/2."""
    x, w = np.polynomial.legendre.leggauss(n_steps)
    return 0.5 * (1.0 + x), 0.5 * w


def integrated_gradients(waves, *model, n_steps: int = 50, internal_batch: int = 8):
    """captum.attr.IntegratedGradients defaults (zero baseline).  Scaled inputs are evaluated
    step-major in chunks of ``internal_batch`` path points (mathematically identical)."""
    alphas, steps = gauss_legendre(n_steps)
    B = waves.shape[0]
    total = torch.zeros_like(waves)
    for s0 in range(0, n_steps, internal_batch):
        a = torch.tensor(alphas[s0:s0 + internal_batch], dtype=waves.dtype)
        scaled = (a[:, None, None] * waves[None]).reshape(-1, waves.shape[-1])        # step-major
        g = input_gradient(scaled, *model).view(len(a), B, -1)
        total += (g * torch.tensor(steps[s0:s0 + internal_batch], dtype=waves.dtype)[:, None, None]).sum(0)
    return total * waves


def time_mask(attr: torch.Tensor) -> torch.Tensor:
    """captum_saliency.py:136-139 -- |attr| / (max|attr| + 1e-8), per clip."""
    a = attr.abs()
    return a / (a.amax(dim=-1, keepdim=True) + 1e-8)
