"""Oracle: HiFi-GAN V1 generator.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED: the reference only holds a handle (hifigan.py:106-110, 180:
``HIFIGAN.from_hparams(...).decode_batch(mel)``); the generator lives in SpeechBrain, which is
absent.  This restates the published V1 generator (Kong et al. 2020, ``models.py::Generator`` with
``ResBlock1``; weight-norm folded) with ``torch.nn.functional`` ops.  SpeechBrain wraps the same
graph; its conv wrappers' padding mode cannot be verified offline, so it is a parameter here
(``"zeros"`` = the published model)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _conv1d(x, w, b, dilation=1, padding_mode="zeros"):
    k = w.shape[-1]
    pad = (k * dilation - dilation) // 2
    if padding_mode == "zeros":
        return F.conv1d(x, w, b, padding=pad, dilation=dilation)
    return F.conv1d(F.pad(x, (pad, pad), mode=padding_mode), w, b, dilation=dilation)


def resblock1(x, sd, p, dilations, slope, padding_mode):
    """ResBlock1: for each dilation d: x += conv2(lrelu(conv1_d(lrelu(x))))."""
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, slope)
        xt = _conv1d(xt, sd[p + f"convs1.{i}.weight"], sd[p + f"convs1.{i}.bias"], d, padding_mode)
        xt = F.leaky_relu(xt, slope)
        xt = _conv1d(xt, sd[p + f"convs2.{i}.weight"], sd[p + f"convs2.{i}.bias"], 1, padding_mode)
        x = xt + x
    return x


def generator(mel: torch.Tensor, sd, cfg, padding_mode: str = "zeros", inference_padding: int = 0) -> torch.Tensor:
    """``mel [B, 80, T] -> wav [B, 1, (T + 2 * inference_padding) * 256]``.  ``inference_padding``: frames replicated on both
    sides first (``F.pad(c, (p, p), "replicate")`` of the SpeechBrain / Coqui generator's ``inference``)."""
    if inference_padding:
        mel = F.pad(mel, (inference_padding, inference_padding), mode="replicate")
    x = _conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], 1, padding_mode)
    nk = len(cfg.resblock_kernel_sizes)
    for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, cfg.leaky_slope)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=r, padding=(k - r) // 2)
        xs = None
        for j in range(nk):
            y = resblock1(x, sd, f"resblocks.{i * nk + j}.", cfg.resblock_dilations, cfg.leaky_slope, padding_mode)
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)                      # default slope 0.01, as published
    x = _conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], 1, padding_mode)
    return torch.tanh(x)


def align_waveforms(ref_wav: torch.Tensor, deg_wav: torch.Tensor):
    """hifigan.py:113-136 -- full cross-correlation by conv1d, argmax shift, trim to common length."""
    ref_wav = ref_wav.view(1, 1, -1)
    deg_wav = deg_wav.view(1, 1, -1)
    padding = deg_wav.shape[-1]
    cc = F.conv1d(F.pad(ref_wav, (padding, padding)), deg_wav)
    shift = int(torch.argmax(cc).item()) - padding
    if shift > 0:
        ref_a = ref_wav[..., shift:]
        deg_a = deg_wav[..., : ref_a.shape[-1]]
    else:
        deg_a = deg_wav[..., -shift:]
        ref_a = ref_wav[..., : deg_a.shape[-1]]
    n = min(ref_a.shape[-1], deg_a.shape[-1])
    return ref_a[..., :n], deg_a[..., :n]


def band_swap_hann(s_ref: torch.Tensor, s_voc: torch.Tensor, band_hz: int = 1000, f_max: int = 8000):
    """hifigan.py:190-228 -- Hann-1024 / hop-256 STFT of the aligned original and vocoded signals; for every
    1 kHz band the complex bins with ``start <= f < end`` on ``linspace(0, 8000, 513)`` are taken from the vocoded
    spectrogram; ``torch.istft`` (no ``length``).  Returns ``(waves [n_bands, L'], leakage [n_bands])``."""
    window = torch.hann_window(1024)
    kw = dict(n_fft=1024, hop_length=256, win_length=1024, window=window)
    X_r = torch.stft(s_ref, return_complex=True, **kw)
    X_v = torch.stft(s_voc, return_complex=True, **kw)
    freqs = torch.linspace(0, 8000, X_r.shape[0])
    waves, leak = [], []
    for start in range(0, f_max, band_hz):
        mask = (freqs >= start) & (freqs < start + band_hz)
        X_c = X_r.clone()
        X_c[mask, :] = X_v[mask, :]
        leak.append(torch.mean((X_c[~mask].abs() - X_r[~mask].abs()) ** 2).item())
        waves.append(torch.istft(X_c, **kw))
    return torch.stack(waves), torch.tensor(leak)
