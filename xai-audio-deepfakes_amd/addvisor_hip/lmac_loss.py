"""LMAC training loss on the HIP path, forward AND backward to the mask.

Reference: ``LMACLoss.loss_function`` (loss_function.py:32-66) and the backward half of the training step
(train_addvisor.py:374-378): ``loss.backward()`` runs through ISTFT x2 and the frozen wav2vec2 + logreg x2 down
to the mask the U-Net produced.  Here that whole chain is one ``torch.autograd.Function``:

  forward   mask -> masked ISTFT x2 (advh_istft_masked) -> ONE 2B-clip embedder pass with saves
            (EmbedderGrad.forward) -> BCE-with-logits x2, L1
  backward  per-clip seeds (sigmoid(logit) - target) / B -> EmbedderGrad.backward (vector-Jacobian product to the
            two resynthesised waveforms) -> advh_istft_masked_bwd x2 (ISTFT adjoint + mask chain rule)

The embedder is frozen, so the backward is dgrad only (no weight gradients).  The three loss terms come back as
one tensor with a grad_fn; ``total = sum(softplus(w_raw) * terms)`` is left to torch so ``w_raw`` trains as in
the reference.  The waveform gradients are computed eagerly inside ``forward`` with unit upstream weights (the
terms are linear in them) and scaled in ``backward``; they are skipped when the mask does not require grad.
"""
from __future__ import annotations

import torch

from . import ops
from .embedder_grad import EmbedderGrad


class _LMACTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, mag, phase, class_pred, eg: EmbedderGrad, length: int, hop: int, win: int, loss_scale: float):
        B = mask.shape[0]
        need = ctx.needs_input_grad[0]
        m = mask.detach().contiguous()
        w_in, w_out = ops.istft_masked(mag, phase, m, length, domain="linear", hop=hop, win=win)
        logits, _ = eg.forward(torch.cat([w_in, w_out], 0))
        l_rel, l_irr = logits[:B].reshape(-1), logits[B:].reshape(-1)
        cp = class_pred.reshape(-1)
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        terms = torch.stack([bce(l_rel, cp), bce(l_irr, 1 - cp), m.abs().mean()])
        if need:
            seed = torch.cat([torch.sigmoid(l_rel) - cp, torch.sigmoid(l_irr) - (1 - cp)]) / B
            g = eg.backward(loss_scale, seed=seed)                          # dL/d wave, [2B, length]
            # fp16 gradients between the dgrad GEMMs: an overflow shows up as inf / NaN here.  Back the power-of-two scale
            # off (exact) and redo the backward rather than hand a poisoned gradient to the optimiser (GradScaler's rule).
            tries = 0
            while not bool(torch.isfinite(g).all()) and tries < 6:
                loss_scale *= 1.0 / 16.0
                tries += 1
                g = eg.backward(loss_scale, seed=seed)
            if tries and not bool(torch.isfinite(g).all()):
                raise FloatingPointError("LMAC loss backward: non-finite input gradient at every loss scale down to "
                                         f"{loss_scale:g}")
            g_in = ops.istft_masked_bwd(g[:B], mag, phase, m, 0, domain="linear", hop=hop, win=win)
            g_out = ops.istft_masked_bwd(g[B:], mag, phase, m, 1, domain="linear", hop=hop, win=win)
            ctx.save_for_backward(g_in, g_out, m)
        return terms

    @staticmethod
    def backward(ctx, go):
        g_in, g_out, m = ctx.saved_tensors
        grad = go[0] * g_in + go[1] * g_out + (go[2] / m.numel()) * torch.sign(m)
        return grad, None, None, None, None, None, None, None, None


def lmac_terms(mask: torch.Tensor, mag: torch.Tensor, phase: torch.Tensor, class_pred: torch.Tensor, eg: EmbedderGrad,
               length: int, hop: int = 322, win: int = 644, loss_scale: float = 4096.0) -> torch.Tensor:
    """``[l_in, l_out, l1]`` (loss_function.py:54-59) for ``mask [B, Fm, Tm]`` (the U-Net crop; bins outside it are
    mask 0), differentiable w.r.t. ``mask``."""
    if mask.dim() != 3:
        raise ValueError("mask must be [B, Fm, Tm]")
    dev = eg.emb.dev
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    return _LMACTerms.apply(f32(mask), f32(mag), f32(phase), f32(class_pred), eg, int(length), hop, win, loss_scale)
