"""Oracle: normaliser, STFT, ISTFT, mask application.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from __future__ import annotations

import warnings

import torch
import torch.nn.functional as F


def zero_mean_unit_var_norm(x: torch.Tensor) -> torch.Tensor:
    """classifier_embedder.py:59-63 -- row-wise, *unbiased* std, eps added to the std."""
    mean = x.mean(dim=-1, keepdim=True)
    std = x.std(dim=-1, keepdim=True)
    return (x - mean) / (std + 1e-7)


def pad_or_crop(w: torch.Tensor, length: int) -> torch.Tensor:
    """audioprocessor.py:83-98 -- zero-pad the tail or crop to ``audio_length * sr``."""
    cur = w.shape[-1]
    if cur < length:
        return F.pad(w, (0, length - cur))
    return w[..., :length]


def compute_stft(w: torch.Tensor, n_fft=1024, hop=322, win=644, audio_length=5, sr=16000):
    """audioprocessor.py:82-112 -- torch.stft with NO window argument: a rectangular ``win``-long
    window zero-padded (centred) to ``n_fft``, center=True, reflect padding, onesided."""
    if w.dim() not in (1, 2):
        raise ValueError("waveform must be 1D (single) or 2D (batched waveforms)")
    w = pad_or_crop(w, int(audio_length * sr))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = torch.stft(w, n_fft=n_fft, hop_length=hop, win_length=win, return_complex=True)
    return X, X.abs(), X.angle()


def compute_invert_stft(spec: torch.Tensor, n_fft=1024, hop=322, win=644, audio_length=5, sr=16000):
    """audioprocessor.py:117-131 -- torch.istft, rectangular window, ``length=audio_length*sr``."""
    if not torch.is_complex(spec):
        raise ValueError("ISTFT expects complex input!")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.istft(spec, n_fft=n_fft, hop_length=hop, win_length=win,
                           length=int(audio_length * sr))


def embed_mask(mask: torch.Tensor, F_full: int, T_full: int) -> torch.Tensor:
    """SURVEY.md D2/D3 adapter (the build's definition; the reference has no runnable one):
    a U-Net mask ``[B, F', T']`` (F'=512, T'=4*floor(T/4)) is embedded into the full ``[B, F, T]``
    grid with ZEROS outside the crop, so mask-out (1-mask) keeps those bins untouched."""
    B, Fm, Tm = mask.shape
    full = mask.new_zeros(B, F_full, T_full)
    full[:, :Fm, :Tm] = mask
    return full


def apply_mask(mask_full: torch.Tensor, mag: torch.Tensor, phase: torch.Tensor, domain: str):
    """Mask-in / mask-out complex spectrograms.

    ``domain="linear"``: loss_function.py:36-45  (m*|X|, (1-m)*|X|) * exp(j*phase).
    ``domain="log1p"``:  LMAC_metrics.py:136-153 expm1(m*log1p|X|), expm1((1-m)*log1p|X|)."""
    if domain == "linear":
        rel, irr = mask_full * mag, (1 - mask_full) * mag
    elif domain == "log1p":
        lm = torch.log1p(mag)
        rel, irr = torch.expm1(mask_full * lm), torch.expm1((1 - mask_full) * lm)
    else:
        raise ValueError(domain)
    ph = torch.exp(1j * phase)
    return rel * ph, irr * ph


def mel_filterbank_slaney(n_mels=80, n_fft=1024, sr=16000, f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """Slaney-scale, slaney-normalised triangular filterbank ``[n_fft//2+1, n_mels]``
    (the ``norm="slaney", mel_scale="slaney"`` arguments of hifigan.py:163-178; algorithm as
    published in Slaney's Auditory Toolbox / librosa.filters.mel)."""
    def hz_to_mel(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz, logstep = 1000.0, torch.log(torch.tensor(6.4, dtype=torch.float64)) / 27.0
        min_log_mel = min_log_hz / f_sp
        return torch.where(f >= min_log_hz, min_log_mel + torch.log(f.clamp(min=1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        f_sp = 200.0 / 3
        min_log_hz, logstep = 1000.0, torch.log(torch.tensor(6.4, dtype=torch.float64)) / 27.0
        min_log_mel = min_log_hz / f_sp
        return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = torch.linspace(0, sr // 2, n_fft // 2 + 1, dtype=torch.float64)
    m_pts = torch.linspace(float(hz_to_mel(f_min)), float(hz_to_mel(f_max)), n_mels + 2, dtype=torch.float64)
    f_pts = mel_to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return (fb * enorm.unsqueeze(0)).to(torch.float32)


def mel_spectrogram(audio: torch.Tensor, sr=16000, hop=256, win=1024, n_mels=80, n_fft=1024,
                    f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """hifigan.py:163-178 -- SpeechBrain ``mel_spectogram(power=1, normalized=False,
    norm="slaney", mel_scale="slaney", compression=True)``: Hann-1024 / hop-256 centred STFT
    magnitude -> slaney mel -> log(clamp(., 1e-5)).  ``[.., L] -> [.., n_mels, 1 + L//hop]``."""
    window = torch.hann_window(win, periodic=True, dtype=audio.dtype)
    X = torch.stft(audio, n_fft=n_fft, hop_length=hop, win_length=win, window=window,
                   center=True, pad_mode="reflect", return_complex=True)
    fb = mel_filterbank_slaney(n_mels, n_fft, sr, f_min, f_max).to(audio.dtype)
    mel = torch.matmul(X.abs().transpose(-1, -2), fb).transpose(-1, -2)
    return torch.log(torch.clamp(mel, min=1e-5))


def band_swap_rect(w_real: torch.Tensor, w_vocoded: torch.Tensor, audio_length=5, band_hz: int = 1000, f_max: int = 8000):
    """train_logReg_swapping.py:64-81 -- AudioProcessor STFTs (rectangular 644 window, hop 322) of one real and one
    vocoded clip; per 1 kHz band the bins ``start <= f < end`` on ``linspace(0, 8000, 513)`` come from the vocoded one;
    ``compute_invert_stft``.  ``[L] , [L] -> [n_bands, audio_length * 16000]``."""
    X_o = compute_stft(w_real, audio_length=audio_length)[0]
    X_v = compute_stft(w_vocoded, audio_length=audio_length)[0]
    freqs = torch.linspace(0, 16000 / 2, X_o.shape[0])
    out = []
    for start in range(0, f_max, band_hz):
        mask = (freqs >= start) & (freqs < start + band_hz)
        X_c = X_o.clone()
        X_c[mask, :] = X_v[mask, :]
        out.append(compute_invert_stft(X_c, audio_length=audio_length))
    return torch.stack(out)
