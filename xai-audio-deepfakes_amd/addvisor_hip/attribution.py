"""Gradient attributions of the waveform -> logit classifier on the HIP backward path: the semantics of
``captum.attr.Saliency / InputXGradient / IntegratedGradients`` as the reference calls them
(captum_saliency.py:116-118, 131-143; Captum defaults: ``abs=True``; IG ``n_steps=50``,
``method="gausslegendre"``, zero baseline, ``multiply_by_inputs=True``, scaled inputs concatenated step-major).

IntegratedGradients is path-batched: the ``n_steps * B`` interpolation points are pushed through one
forward + dgrad-only backward in chunks of ``internal_batch_size`` rows (whole steps per chunk).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .embedder import HipEmbedder
from .embedder_grad import EmbedderGrad


def _st():
    return torch.cuda.current_stream().cuda_stream


def gauss_legendre(n_steps: int) -> Tuple[np.ndarray, np.ndarray]:
    """Captum's ``gausslegendre`` rule: alphas = (1 + x) / 2, step sizes = w / 2."""
    x, w = np.polynomial.legendre.leggauss(n_steps)
    return 0.5 * (1.0 + x), 0.5 * w


class HipAttribution:
    def __init__(self, emb: HipEmbedder, loss_scale: float = 4096.0, precision: Optional[str] = None):
        """``precision``: None = the embedder's (an fp32-class embedder gives the fp32-class gradient chain, the reference's
        fp32 autograd class); "f16" = the fp16-operand chain."""
        self.emb, self.eg, self.loss_scale = emb, EmbedderGrad(emb, precision), loss_scale
        self.precision = self.eg.precision

    def _prep(self, waves: torch.Tensor) -> torch.Tensor:
        if waves.dim() == 1:
            waves = waves[None]
        return waves.to(self.emb.dev, torch.float32).contiguous()

    def input_gradient(self, waves: torch.Tensor) -> torch.Tensor:
        """d logit / d wave, ``[B, L]`` fp32."""
        x = self._prep(waves)
        self.eg.forward(x)
        return self.eg.backward(self.loss_scale)

    def _finalize(self, g, x, mode):
        out = torch.empty_like(g)
        _lib.check(_lib.lib().advh_attr_finalize(g.data_ptr(), x.data_ptr(), out.data_ptr(), mode, g.numel(), _st()), "advh_attr_finalize")
        # the planes between the dgrad GEMMs have fp16's exponent range: an overflow (|scaled gradient| > 65504 somewhere in the
        # chain) surfaces as inf / NaN in the input gradient and in every sum over path points.  One flag read per attribution:
        # raise instead of handing back a poisoned attribution map.
        finite = bool(torch.isfinite(out).all())             # synchronises: every kernel of the chain has run
        _lib.check_overflow("attribution")                   # fp32-class chain: saturated planes raise SplitRangeError (a FloatingPointError)
        if not finite:
            raise FloatingPointError(f"non-finite attribution: the gradient chain overflowed at loss_scale={self.loss_scale:g} "
                                     "(lower HipAttribution.loss_scale by a power of two)")
        return out

    def saliency(self, waves):
        x = self._prep(waves)
        return self._finalize(self.input_gradient(x), x, 0)

    def input_x_gradient(self, waves):
        x = self._prep(waves)
        return self._finalize(self.input_gradient(x), x, 1)

    def integrated_gradients(self, waves, n_steps: int = 50, internal_batch_size: Optional[int] = None):
        x = self._prep(waves)
        B, L = x.shape
        alphas, steps = gauss_legendre(n_steps)
        per = max(1, (internal_batch_size or 128) // B)            # whole steps per chunk
        total = torch.zeros_like(x)
        lib = _lib.lib()
        per = min(per, n_steps)
        npad = -(-n_steps // per) * per                              # every chunk has the same shape: one workspace
        alphas = np.concatenate([alphas, np.full(npad - n_steps, alphas[-1])])
        steps = np.concatenate([steps, np.zeros(npad - n_steps)])    # padding steps carry zero weight
        a_all = torch.tensor(np.repeat(alphas, B), dtype=torch.float32, device=x.device)
        w_all = torch.tensor(np.repeat(steps, B), dtype=torch.float32, device=x.device)
        scaled = torch.empty((per * B, L), dtype=torch.float32, device=x.device)
        for s0 in range(0, npad, per):
            ns = per
            a = a_all[s0 * B:(s0 + ns) * B]
            _lib.check(lib.advh_scale_rows(x.data_ptr(), B, a.data_ptr(), scaled.data_ptr(), ns * B, L, 0, _st()), "advh_scale_rows")
            self.eg.forward(scaled)
            g = self.eg.backward(self.loss_scale)                    # [ns*B, L], step-major
            for k in range(ns):
                wk = w_all[(s0 + k) * B:(s0 + k + 1) * B]
                _lib.check(lib.advh_scale_rows(g[k * B:(k + 1) * B].data_ptr(), B, wk.data_ptr(), total.data_ptr(), B, L, 1, _st()),
                           "advh_scale_rows")
        return self._finalize(total, x, 1)

    def time_mask(self, attr: torch.Tensor, waves: Optional[torch.Tensor] = None):
        """``|attr| / (max|attr| + 1e-8)`` per clip and, with ``waves``, the relevant / irrelevant waveforms
        (captum_saliency.py:136-143)."""
        attr = attr.contiguous()
        B, L = attr.shape
        mask = torch.empty_like(attr)
        if waves is None:
            _lib.check(_lib.lib().advh_time_mask(attr.data_ptr(), mask.data_ptr(), None, None, None, B, L, _st()), "advh_time_mask")
            return mask
        x = self._prep(waves)
        win, wout = torch.empty_like(x), torch.empty_like(x)
        _lib.check(_lib.lib().advh_time_mask(attr.data_ptr(), mask.data_ptr(), win.data_ptr(), wout.data_ptr(), x.data_ptr(), B, L, _st()),
                   "advh_time_mask")
        return mask, win, wout
