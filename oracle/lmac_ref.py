"""Oracle: LMAC metrics, LMAC loss forward, and the whole explanation pipeline.
TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import signal_ref as S
from . import unet_ref as U
from . import wav2vec2_ref as W

EPS = 1e-10  # LMAC_metrics.py:28


# ---- LMAC_metrics.py:31-73 (duplicated in captum_saliency.py:68-81); all take [N,1] probabilities
def compute_fidelity(theta_out, predictions, threshold=0.5):
    """LMAC_metrics.py:31-38."""
    return ((predictions > threshold).long() == (theta_out > threshold).long()).float()


def get_score_for_predicted_class(p):
    """LMAC_metrics.py:43-45."""
    pred = (p > 0.5).float()
    return pred * p + (1 - pred) * (1 - p)


def compute_faithfulness(predictions, predictions_masked):
    """LMAC_metrics.py:48-52."""
    return ((predictions - predictions_masked) * torch.sign(predictions - 0.5)).squeeze(dim=1)


def compute_AD(theta_out, predictions):
    """LMAC_metrics.py:55-59."""
    pc = get_score_for_predicted_class(predictions.squeeze(1))
    oc = get_score_for_predicted_class(theta_out.squeeze(1))
    return (F.relu(pc - oc) / (pc + EPS)) * 100


def compute_AI(theta_out, predictions):
    """LMAC_metrics.py:62-66."""
    pc = get_score_for_predicted_class(predictions.squeeze(1))
    oc = get_score_for_predicted_class(theta_out.squeeze(1))
    return (oc > pc).float() * 100


def compute_AG(theta_out, predictions):
    """LMAC_metrics.py:69-73."""
    pc = get_score_for_predicted_class(predictions.squeeze(1))
    oc = get_score_for_predicted_class(theta_out.squeeze(1))
    return (F.relu(oc - pc) / (1 - pc + EPS)) * 100


def lmac_summary(predictions, theta_out, masked_predictions) -> Dict[str, float]:
    """The five means printed at LMAC_metrics.py:164-172 (fp32 means, like the reference)."""
    return {
        "faithfulness": compute_faithfulness(predictions, masked_predictions).mean().item(),
        "fidelity": compute_fidelity(theta_out, predictions).float().mean().item(),
        "AD": compute_AD(theta_out, predictions).mean().item(),
        "AI": compute_AI(theta_out, predictions).mean().item(),
        "AG": compute_AG(theta_out, predictions).mean().item(),
    }


# ---- loss_function.py:32-66
def lmac_loss(xhat, X_stft_power, X_stft_phase, class_pred, w_raw, emb_sd, emb_cfg, coef, intercept,
              audio_length=5):
    """LMACLoss.loss_function forward as written (loss_function.py:32-66): ``xhat [B,1,F',T']``
    must broadcast against ``X[:, :F', :]``, i.e. F'=513 and T'=T for the code to run."""
    xhat = xhat.squeeze(1)
    Tmax = xhat.shape[1]
    mag, ph = X_stft_power[:, :Tmax, :], X_stft_phase[:, :Tmax, :]
    rel = (xhat * mag) * torch.exp(1j * ph)
    irr = ((1 - xhat) * mag) * torch.exp(1j * ph)
    w_rel = S.compute_invert_stft(rel, audio_length=audio_length)
    w_irr = S.compute_invert_stft(irr, audio_length=audio_length)
    f_rel = W.extract_features(w_rel, emb_sd, emb_cfg)
    f_irr = W.extract_features(w_irr, emb_sd, emb_cfg)
    l_rel, _ = W.logreg(torch.mean(f_rel.squeeze(0), dim=1), coef, intercept)
    l_irr, _ = W.logreg(torch.mean(f_irr.squeeze(0), dim=1), coef, intercept)
    l_in = F.binary_cross_entropy_with_logits(l_rel, class_pred)
    l_out = F.binary_cross_entropy_with_logits(l_irr, 1 - class_pred)
    reg_l1 = xhat.abs().mean()
    losses = torch.stack([l_in, l_out, reg_l1])
    w = F.softplus(torch.as_tensor(w_raw, dtype=torch.float32))
    return torch.sum(w * losses), losses, w


# ---- the unit of work of SURVEY.md §8d: one explanation per clip
def explain(waves: torch.Tensor, emb_sd, emb_cfg, coef, intercept, unet_sd, audio_length=4,
            domain="log1p", bn_batch=False, vocoder=None) -> Dict[str, torch.Tensor]:
    """LMAC_metrics.py:117-157 with the D1-D5 resolutions of SURVEY.md §2.3:
    STFT -> classifier(clean) -> U-Net(mask) -> mask-in / mask-out resynthesis -> classifier x2.
    ``vocoder = (hifigan_sd, hifigan_cfg)``: the north-star variant "masked spectrogram -> HiFi-GAN vocoder -> classifier
    re-forward": both resyntheses are re-rendered through the mel front end (hifigan.py:163-178) and the V1 generator
    (hifigan.py:180), cropped / zero-padded to the clip length, before the classifier sees them."""
    X, mag, phase = S.compute_stft(waves, audio_length=audio_length)
    _, p_clean = W.classify(S.pad_or_crop(waves, int(audio_length * 16000)), emb_sd, emb_cfg, coef, intercept)
    mask = U.unet_forward(U.crop_for_unet(mag), unet_sd, bn_batch=bn_batch)[:, 0]
    mfull = S.embed_mask(mask, mag.shape[1], mag.shape[2])
    rel, irr = S.apply_mask(mfull, mag, phase, domain)
    w_rel = S.compute_invert_stft(rel, audio_length=audio_length)
    w_irr = S.compute_invert_stft(irr, audio_length=audio_length)
    if vocoder is not None:
        from . import hifigan_ref as H
        hsd, hcfg = vocoder
        L = int(audio_length * 16000)
        w_rel, w_irr = (S.pad_or_crop(H.generator(S.mel_spectrogram(w), hsd, hcfg)[:, 0], L) for w in (w_rel, w_irr))
    _, p_in = W.classify(w_rel, emb_sd, emb_cfg, coef, intercept)
    _, p_out = W.classify(w_irr, emb_sd, emb_cfg, coef, intercept)
    return dict(mag=mag, phase=phase, mask=mask, wave_in=w_rel, wave_out=w_irr,
                predictions=p_clean, theta_out=p_in, masked_predictions=p_out)
