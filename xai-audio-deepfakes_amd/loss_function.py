"""Drop-in for the reference's ``loss_function`` module (loss_function.py:1-77): LMAC loss on the HIP path,
forward and backward.  ``total_loss.backward()`` (train_addvisor.py:376) delivers gradients to ``w_raw`` (torch)
and to ``xhat`` -- through the ISTFT adjoint and the frozen embedder's input-gradient chain, both HIP
(addvisor_hip/lmac_loss.py) -- so the reference's training loop drives a mask decoder unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from addvisor_hip import ops as _ops, runtime as _rt
from audioprocessor import AudioProcessor
from classifier_embedder import TorchLogReg  # noqa: F401  (the reference module exposes it, loss_function.py:15)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
audio_processor = AudioProcessor()


class LMACLoss(nn.Module):
    def __init__(self, reg_w_tv=0.00, loss_scale=4096.0):
        """``loss_scale``: initial power-of-two scale of the gradients inside the frozen embedder's backward (backed off on
        overflow and kept, grown back after clean steps: addvisor_hip/lmac_loss.LossScaler); the reference's signature is
        ``LMACLoss(reg_w_tv=0.0)``."""
        super().__init__()
        self.reg_w_tv = reg_w_tv
        from addvisor_hip.lmac_loss import LossScaler
        self.scaler = LossScaler(float(loss_scale))
        self.w_raw = nn.Parameter(torch.tensor([3.0, 0.5, 3.0], requires_grad=True))     # loss_function.py:24

    @property
    def loss_scale(self) -> float:
        return self.scaler.scale

    @property
    def w(self):
        return F.softplus(self.w_raw)

    def loss_function(self, xhat, X_stft_power, X_stft_phase, class_pred):
        """loss_function.py:32-66.  ``xhat [B,1,F',T']`` is embedded into the 513 x T grid with zeros
        outside; linear masking; two resyntheses; two classifier passes."""
        ap = audio_processor
        L = int(ap.audio_length * ap.sampling_rate)
        m = xhat.squeeze(1).to(device, torch.float32)
        mag = X_stft_power.to(device, torch.float32).contiguous()
        ph = X_stft_phase.to(device, torch.float32).contiguous()
        cp = class_pred.to(device, torch.float32)
        if torch.is_grad_enabled() and m.requires_grad:
            from addvisor_hip.lmac_loss import lmac_terms
            losses = lmac_terms(m, mag, ph, cp, _rt.hip_embedder_grad(), L, hop=ap.hop_length, win=ap.win_length,
                                loss_scale=self.scaler)
        else:
            w_in, w_out = _ops.istft_masked(mag, ph, m.detach(), L, domain="linear", hop=ap.hop_length, win=ap.win_length)
            emb = _rt.hip_embedder()
            _, l_rel, _ = emb.forward(w_in, L, want_hidden=False)
            _, l_irr, _ = emb.forward(w_out, L, want_hidden=False)
            l_in = F.binary_cross_entropy_with_logits(l_rel, cp)
            l_out = F.binary_cross_entropy_with_logits(l_irr, 1 - cp)
            losses = torch.stack([l_in, l_out, m.abs().mean()])
        w = self.w.to(device)
        return torch.sum(w * losses), losses, self.w
