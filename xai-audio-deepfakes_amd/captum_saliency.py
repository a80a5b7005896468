"""Drop-in for the reference's ``captum_saliency`` module (captum_saliency.py:1-219): ``Wav2vec2LogReg``,
the metric helpers and ``compute_camptum_saliency_metrics``, on the HIP forward + dgrad-only backward.
Nothing runs at import (the reference executes the whole evaluation at import, captum_saliency.py:215-219)."""
import os

import torch
import torch.nn as nn

from addvisor_hip import pipeline as _P, runtime as _rt
from audioprocessor import AudioProcessor
from captum.attr import InputXGradient, IntegratedGradients, Saliency  # noqa: F401
from classifier_embedder import TorchLogReg  # noqa: F401  (name kept for callers of the reference module)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
audioprocessor = AudioProcessor()


@torch.no_grad()
def compute_fidelity(theta_out, predictions, threshold=0.5):
    """captum_saliency.py:68-75."""
    return ((predictions > threshold).long() == (theta_out > threshold).long()).float()


def compute_faithfulness(predictions, predictions_masked):
    """captum_saliency.py:78-81."""
    return ((predictions - predictions_masked) * torch.sign(predictions - 0.5)).squeeze(dim=1)


class Wav2vec2LogReg(nn.Module):
    """captum_saliency.py:84-100: waveform -> logit, pooled over time per example."""

    def __init__(self, audioprocessor, logReg):
        super().__init__()
        self.ap = audioprocessor
        self.logReg = logReg
        self._att = None

    def forward(self, waveform):
        logits, _ = self.ap.classify(waveform)
        return logits

    def hip_attribution(self):
        if self._att is None:
            from addvisor_hip.attribution import HipAttribution
            self._att = HipAttribution(_rt.hip_embedder())
        return self._att


def extract_wavs(metadata):
    """captum_saliency.py:103-109."""
    audio_files = []
    with open(metadata, "r") as f:
        for line in f:
            audio_files.append(line.strip().split(",")[0])
    return audio_files


def explain_waves(model, waves, method="input_x_gradient", n_steps=50):
    """Loop body of compute_camptum_saliency_metrics (captum_saliency.py:125-192) for a batch ``[B, L]``:
    attribution -> |attr|/max time mask -> wave*mask, wave*(1-mask) -> three classifier passes.
    Returns ``(predictions, theta_out, masked_predictions)``, each ``[B,1]``."""
    att = model.hip_attribution()
    x = waves.to(device, torch.float32)
    attr = {"saliency": att.saliency, "input_x_gradient": att.input_x_gradient,
            "integrated_gradients": lambda w: att.integrated_gradients(w, n_steps=n_steps)}[method](x)
    _, w_rel, w_irr = att.time_mask(attr, x)
    emb = _rt.hip_embedder()
    B = x.shape[0]
    _, _, p = emb.forward(torch.cat([x, w_rel, w_irr], 0), want_hidden=False)
    return p[:B], p[B:2 * B], p[2 * B:]


def compute_camptum_saliency_metrics(model, metadata_path, target_class=None, root="LJSpeech_vocoded",
                                     method="input_x_gradient", batch_size=8):
    """captum_saliency.py:112-212 (name kept as in the reference); prints faithfulness and fidelity."""
    model.eval()
    wav_paths = extract_wavs(metadata_path)
    print(f"computing saliency for {len(wav_paths)} files")
    preds, thetas, masked = [], [], []
    for i in range(0, len(wav_paths), batch_size):
        waves = torch.stack([audioprocessor.load_audio(os.path.join(root, p))[0] for p in wav_paths[i:i + batch_size]])
        p, t, o = explain_waves(model, waves, method)
        preds.append(p), thetas.append(t), masked.append(o)
    predictions, theta_out, masked_predictions = torch.cat(preds), torch.cat(thetas), torch.cat(masked)
    m = _P.lmac_metrics(predictions, theta_out, masked_predictions)
    print(f"faithfulness : {m['faithfulness']:.2f}")
    print(f"fidelity: {m['fidelity']:.2f}")
    counter = int((theta_out[-batch_size:] >= 0.5).sum().item())
    print(f"number of relevant masks classified as manipulated: {counter} out of {min(batch_size, len(wav_paths))}")
    return None
