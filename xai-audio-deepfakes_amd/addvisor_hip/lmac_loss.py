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

from . import _lib, ops
from .embedder_grad import EmbedderGrad


class LossScaler:
    """Power-of-two scale of the gradients inside the frozen embedder's backward, GradScaler style: backed off by 1/16 when
    the input gradient comes back non-finite (the planes between the dgrad GEMMs have fp16's exponent range in both
    precision modes) and KEPT for the following steps, doubled again after ``growth_interval`` clean steps up to the
    initial value.  Powers of two are exact, so the scale never changes a finite result."""

    def __init__(self, scale: float = 4096.0, growth_interval: int = 200, min_scale: float = 2.0 ** -20):
        self.initial = self.scale = float(scale)
        self.growth_interval, self.min_scale = int(growth_interval), float(min_scale)
        self.good_steps = 0

    def backoff(self) -> bool:
        self.good_steps = 0
        if self.scale * (1.0 / 16.0) < self.min_scale:
            return False
        self.scale *= 1.0 / 16.0
        return True

    def good(self):
        self.good_steps += 1
        if self.good_steps >= self.growth_interval and self.scale < self.initial:
            self.scale, self.good_steps = min(self.initial, self.scale * 2.0), 0


class _LMACTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, mag, phase, class_pred, eg: EmbedderGrad, length: int, hop: int, win: int, scaler: LossScaler):
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
            # One flag read per step (the only host synchronisation of the loss): an overflow shows up as inf / NaN in the input
            # gradient.  Back the power-of-two scale off (exact), keep it for the following steps and redo this backward
            # rather than hand a poisoned gradient to the optimiser.  A NaN that is not an overflow (bad input) fails at once
            # when the forward logits are already non-finite.
            def attempt():
                """dL/d wave at the current scale and whether it is usable.  An overflow shows up as inf / NaN (fp16 chain) or
                as the split format's sticky range flag (fp32-class chain: its planes saturate) -- possibly as a SplitRangeError
                out of a later launch of the same chain."""
                try:
                    gg = eg.backward(scaler.scale, seed=seed)               # [2B, length]
                    flags = torch.stack([torch.isfinite(logits).all(), torch.isfinite(gg).all()]).tolist()   # ONE read for both flags (synchronises)
                    _lib.check_overflow("LMAC loss backward")
                    return gg, flags[0], flags[1]
                except _lib.SplitRangeError:
                    torch.cuda.synchronize()
                    _lib.lib().advh_split_overflow(1)
                    return None, bool(torch.isfinite(logits).all()), False

            g, ok_fwd, ok = attempt()
            if not ok_fwd:
                raise FloatingPointError("LMAC loss: non-finite classifier logits in the forward pass (bad input or weights)")
            while not ok:
                if not scaler.backoff():
                    raise FloatingPointError(f"LMAC loss backward: non-finite input gradient at every loss scale down to {scaler.scale:g}")
                g, _, ok = attempt()
            scaler.good()
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
               length: int, hop: int = 322, win: int = 644, loss_scale=4096.0) -> torch.Tensor:
    """``[l_in, l_out, l1]`` (loss_function.py:54-59) for ``mask [B, Fm, Tm]`` (the U-Net crop; bins outside it are
    mask 0), differentiable w.r.t. ``mask``."""
    if mask.dim() != 3:
        raise ValueError("mask must be [B, Fm, Tm]")
    dev = eg.emb.dev
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    scaler = loss_scale if isinstance(loss_scale, LossScaler) else LossScaler(float(loss_scale))   # a LossScaler persists back-offs across steps
    return _LMACTerms.apply(f32(mask), f32(mag), f32(phase), f32(class_pred), eg, int(length), hop, win, scaler)
