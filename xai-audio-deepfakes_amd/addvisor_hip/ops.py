"""Torch-tensor front ends of the C-ABI entry points.

PyTorch is plumbing here: it owns device memory (caching allocator) and the current HIP stream.
Every function checks shapes / dtypes / contiguity on the host before a raw pointer reaches a
hand-written kernel, launches on ``torch.cuda.current_stream()`` and returns torch tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

NFFT = 1024
NBIN = 513


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA (HIP) tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def stft_forward(wave: torch.Tensor, length: int, hop: int = 322, win: int = 644,
                 window: Optional[torch.Tensor] = None, want_complex: bool = True,
                 want_mag: bool = True, want_phase: bool = True
                 ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """``wave [B, n_in]`` fp32 -> ``X [B,513,T] complex64, |X|, angle(X) [B,513,T]`` (audioprocessor.py:82-112).

    The clip is zero-padded / cropped to ``length`` inside the kernel."""
    _lib.init()
    wave = _req(wave, torch.float32, "wave")
    if wave.dim() != 2:
        raise ValueError("wave must be [B, n]")
    B, n_in = wave.shape
    T = 1 + length // hop
    dev = wave.device
    X = torch.empty((B, NBIN, T, 2), dtype=torch.float32, device=dev) if want_complex else None
    mag = torch.empty((B, NBIN, T), dtype=torch.float32, device=dev) if want_mag else None
    ph = torch.empty((B, NBIN, T), dtype=torch.float32, device=dev) if want_phase else None
    if window is not None:
        window = _req(window, torch.float32, "window")
        if window.numel() != win:
            raise ValueError("window must have `win` elements")
    rc = _lib.lib().advh_stft_forward(wave.data_ptr(), wave.stride(0), n_in, B, length, hop, win, _ptr(window),
                                      _ptr(X), _ptr(mag), _ptr(ph), T, _stream())
    _lib.check(rc, "advh_stft_forward")
    return (torch.view_as_complex(X) if X is not None else None), mag, ph


def istft_masked(mag: torch.Tensor, phase: torch.Tensor, mask: Optional[torch.Tensor], length: int,
                 domain: str = "log1p", want_in: bool = True, want_out: bool = True, hop: int = 322,
                 win: int = 644, window: Optional[torch.Tensor] = None
                 ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Mask-in / mask-out resynthesis, fused: mask application (``"linear"`` loss_function.py:36-45,
    ``"log1p"`` LMAC_metrics.py:136-153, ``"none"``) + polar + ISTFT (audioprocessor.py:117-131).

    ``mask [B, Fm, Tm]`` is the U-Net crop; bins outside it count as mask 0."""
    _lib.init()
    mag = _req(mag, torch.float32, "mag")
    phase = _req(phase, torch.float32, "phase")
    if mag.shape != phase.shape or mag.dim() != 3 or mag.shape[1] != NBIN:
        raise ValueError("mag / phase must be [B, 513, T]")
    B, _, T = mag.shape
    mode = {"none": 0, "linear": 1, "log1p": 2}[domain]
    Fm = Tm = 0
    if mode:
        mask = _req(mask, torch.float32, "mask")
        if mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] > NBIN or mask.shape[2] > T:
            raise ValueError("mask must be [B, Fm<=513, Tm<=T]")
        Fm, Tm = mask.shape[1], mask.shape[2]
    elif want_out:
        raise ValueError("domain='none' has no mask-out signal")
    if T != 1 + length // hop:
        raise ValueError("T does not match length // hop + 1")
    w_in = torch.empty((B, length), dtype=torch.float32, device=mag.device) if want_in else None
    w_out = torch.empty((B, length), dtype=torch.float32, device=mag.device) if want_out else None
    rc = _lib.lib().advh_istft_masked(mag.data_ptr(), phase.data_ptr(), _ptr(mask) if mode else None, Fm, Tm, mode,
                                      _ptr(w_in), _ptr(w_out), length, B, T, length, hop, win, _ptr(window), _stream())
    _lib.check(rc, "advh_istft_masked")
    return w_in, w_out


def istft_masked_c64(spec: torch.Tensor, mask: torch.Tensor, length: int, domain: str = "log1p", want_in: bool = True,
                     want_out: bool = True, hop: int = 322, win: int = 644, window: Optional[torch.Tensor] = None
                     ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """:func:`istft_masked` from the complex spectrogram ``spec [B, 513, T]`` complex64 instead of ``(|X|, angle X)``:
    ``X' = X * g(mask, |X|) / |X|`` -- the same masked signal (loss_function.py:36-45, LMAC_metrics.py:136-153) without the
    atan2 / sincos round trip; both resyntheses come out of one launch that reads the spectrogram from HBM once."""
    _lib.init()
    if not torch.is_complex(spec) or spec.dtype != torch.complex64 or spec.dim() != 3 or spec.shape[1] != NBIN:
        raise ValueError("spec must be complex64 [B, 513, T]")
    sr = torch.view_as_real(spec.contiguous())
    B, _, T = spec.shape
    mode = {"linear": 1, "log1p": 2}[domain]
    mask = _req(mask, torch.float32, "mask")
    if mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] > NBIN or mask.shape[2] > T:
        raise ValueError("mask must be [B, Fm<=513, Tm<=T]")
    if T != 1 + length // hop:
        raise ValueError("T does not match length // hop + 1")
    w_in = torch.empty((B, length), dtype=torch.float32, device=spec.device) if want_in else None
    w_out = torch.empty((B, length), dtype=torch.float32, device=spec.device) if want_out else None
    rc = _lib.lib().advh_istft_masked_c64(sr.data_ptr(), mask.data_ptr(), mask.shape[1], mask.shape[2], mode, _ptr(w_in), _ptr(w_out),
                                          length, B, T, length, hop, win, _ptr(window), _stream())
    _lib.check(rc, "advh_istft_masked_c64")
    return w_in, w_out


def istft_masked_bwd(g_wave: torch.Tensor, mag: torch.Tensor, phase: torch.Tensor, mask: torch.Tensor, which: int,
                     domain: str = "linear", hop: int = 322, win: int = 644,
                     window: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of :func:`istft_masked` for one branch: ``g_wave [B, length]`` = dL/d(resynthesised wave) of the
    mask-in (``which = 0``) or mask-out (``which = 1``) signal -> ``dL/d mask [B, Fm, Tm]``
    (LMACLoss backward, loss_function.py:36-47)."""
    _lib.init()
    mag = _req(mag, torch.float32, "mag")
    phase = _req(phase, torch.float32, "phase")
    mask = _req(mask, torch.float32, "mask")
    g_wave = _req(g_wave, torch.float32, "g_wave")
    B, _, T = mag.shape
    if mag.shape != phase.shape or mag.dim() != 3 or mag.shape[1] != NBIN:
        raise ValueError("mag / phase must be [B, 513, T]")
    if mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] > NBIN or mask.shape[2] > T:
        raise ValueError("mask must be [B, Fm<=513, Tm<=T]")
    if g_wave.dim() != 2 or g_wave.shape[0] != B:
        raise ValueError("g_wave must be [B, length]")
    length = g_wave.shape[1]
    if T != 1 + length // hop:
        raise ValueError("T does not match length // hop + 1")
    mode = {"linear": 1, "log1p": 2}[domain]
    dmask = torch.empty_like(mask)
    rc = _lib.lib().advh_istft_masked_bwd(g_wave.data_ptr(), length, mag.data_ptr(), phase.data_ptr(), mask.data_ptr(),
                                          mask.shape[1], mask.shape[2], mode, int(which), dmask.data_ptr(), B, T, length, hop,
                                          win, _ptr(window), _stream())
    _lib.check(rc, "advh_istft_masked_bwd")
    return dmask


def istft_bandswap(spec_a: torch.Tensor, spec_b: torch.Tensor, length: int, k0: int = 0, kw: int = 64, nbands: int = 8,
                   hop: int = 322, win: int = 644, window: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Band-swap resynthesis (hifigan.py:196-228, train_logReg_swapping.py:64-92): for each band ``z`` the bins
    ``[k0 + z*kw, k0 + (z+1)*kw)`` of ``spec_a`` are replaced by ``spec_b``'s and the spectrogram is inverted;
    ``[nbands, B, length]`` fp32, one launch.  With 513 bins on ``linspace(0, 8000, 513)`` the reference's 1 kHz bands
    ``freqs >= 1000 z and freqs < 1000 (z+1)`` are exactly ``k0 = 0, kw = 64`` (bin 512 = 8 kHz belongs to no band)."""
    _lib.init()
    for sp in (spec_a, spec_b):
        if not torch.is_complex(sp):
            raise ValueError("ISTFT expects complex input!")
        if sp.dtype != torch.complex64 or sp.dim() != 3 or sp.shape[1] != NBIN:
            raise ValueError("spec must be complex64 [B, 513, T]")
    if spec_a.shape != spec_b.shape:
        raise ValueError("the two spectrograms must have the same shape")
    B, _, T = spec_a.shape
    if T != 1 + length // hop:
        raise ValueError("T does not match length // hop + 1")
    if window is not None:
        window = _req(window, torch.float32, "window")
    a = torch.view_as_real(spec_a.contiguous())
    b = torch.view_as_real(spec_b.contiguous())
    out = torch.empty((nbands, B, length), dtype=torch.float32, device=spec_a.device)
    rc = _lib.lib().advh_istft_bandswap(a.data_ptr(), b.data_ptr(), k0, kw, nbands, out.data_ptr(), length, B * length, B, T,
                                        length, hop, win, _ptr(window), _stream())
    _lib.check(rc, "advh_istft_bandswap")
    return out


def istft_complex(spec: torch.Tensor, length: int, hop: int = 322, win: int = 644,
                  window: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``spec [B,513,T] complex64 -> wave [B, length]`` (audioprocessor.py:117-131)."""
    _lib.init()
    if not torch.is_complex(spec):
        raise ValueError("ISTFT expects complex input!")          # audioprocessor.py:118-119
    if spec.dtype != torch.complex64 or spec.dim() != 3 or spec.shape[1] != NBIN:
        raise ValueError("spec must be complex64 [B, 513, T]")
    sr = torch.view_as_real(spec.contiguous())
    B, _, T, _ = sr.shape
    if T != 1 + length // hop:
        raise ValueError("T does not match length // hop + 1")
    out = torch.empty((B, length), dtype=torch.float32, device=spec.device)
    rc = _lib.lib().advh_istft_c64(sr.data_ptr(), out.data_ptr(), length, B, T, length, hop, win, _ptr(window), _stream())
    _lib.check(rc, "advh_istft_c64")
    return out


_MEL_FB = {}


def mel_filterbank(n_mels=80, n_fft=1024, sr=16000, f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """Slaney-scale, slaney-normalised triangular filterbank ``[n_fft//2+1, n_mels]`` (the ``norm="slaney",
    mel_scale="slaney"`` arguments of hifigan.py:171-177), built once on the host in fp64."""
    import numpy as np
    key = (n_mels, n_fft, sr, f_min, f_max)
    if key not in _MEL_FB:
        f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp

        def hz_to_mel(f):
            f = np.asarray(f, dtype=np.float64)
            return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

        def mel_to_hz(m):
            return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

        freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
        f_pts = mel_to_hz(np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2))
        f_diff = np.diff(f_pts)
        slopes = f_pts[None, :] - freqs[:, None]
        fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
        fb = fb * (2.0 / (f_pts[2:] - f_pts[:-2]))[None, :]
        _MEL_FB[key] = torch.from_numpy(fb.astype(np.float32))
    return _MEL_FB[key]


def mel_spectrogram(audio: torch.Tensor, sr=16000, hop=256, win=1024, n_mels=80, f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """``audio [B, L]`` (or ``[L]``) -> log-mel ``[B, n_mels, 1 + L // hop]``: the SpeechBrain ``mel_spectogram``
    call of hifigan.py:163-178 (Hann window, power 1, slaney mel, ``log(clamp(., 1e-5))``)."""
    _lib.init()
    single = audio.dim() == 1
    a = _req(audio[None] if single else audio, torch.float32, "audio")
    B, L = a.shape
    window = torch.hann_window(win, periodic=True, dtype=torch.float32, device=a.device)
    _, mag, _ = stft_forward(a, L, hop, win, window=window, want_complex=False, want_phase=False)
    fb = mel_filterbank(n_mels, NFFT, sr, f_min, f_max).to(a.device)
    T = mag.shape[2]
    out = torch.empty((B, n_mels, T), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().advh_mel_log(mag.data_ptr(), fb.data_ptr(), out.data_ptr(), B, NBIN, T, n_mels, _stream()), "advh_mel_log")
    return out[0] if single else out
