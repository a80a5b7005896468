"""Drop-in for the reference's ``hifigan`` module (hifigan.py:106-136, 163-180): ``hifi_gan.decode_batch``,
``mel_spectogram`` and ``align_waveforms``.  Importing it runs nothing (the reference's dataset loop,
hifigan.py:139-230, is §8(f) rank 2 and not part of the hot path) and fetches nothing: the generator weights
come from ``ADDVISOR_HIFIGAN`` (a ``.pth`` state dict named like ``addvisor_hip.synthetic.hifigan_weights``,
weight-norm folded) or the seeded synthetic generator."""
import os

import torch
import torch.nn.functional as F

from addvisor_hip import ops as _ops, runtime as _rt, synthetic as _syn

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _HifiGan:
    """``hifi_gan.decode_batch(mel [B,80,T]) -> wav [B,1,T*256]`` (no-grad), HiFi-GAN V1 on the HIP kernels."""

    def __init__(self):
        self._net = None

    def _engine(self):
        if self._net is None:
            from addvisor_hip.hifigan import HipHifigan
            cfg = _syn.HifiganConfig()
            path = os.environ.get("ADDVISOR_HIFIGAN")
            sd = torch.load(path, map_location="cpu") if path else _syn.hifigan_weights(cfg)
            self._net = HipHifigan(cfg, sd, _rt.device())
        return self._net

    @torch.no_grad()
    def decode_batch(self, spectrogram, mel_lens=None, hop_len=None):
        return self._engine().decode_batch(spectrogram)

    def decode_spectrogram(self, spectrogram):
        return self.decode_batch(spectrogram[None])[0]


hifi_gan = _HifiGan()


def mel_spectogram(audio, sample_rate=16000, hop_length=256, win_length=1024, n_mels=80, n_fft=1024, f_min=0.0,
                   f_max=8000.0, power=1, normalized=False, min_max_energy_norm=True, norm="slaney",
                   mel_scale="slaney", compression=True):
    """The SpeechBrain call of hifigan.py:163-178 (spelling as in SpeechBrain); returns ``(mel, None)``."""
    if (n_fft, power, normalized, norm, mel_scale, compression) != (1024, 1, False, "slaney", "slaney", True):
        raise NotImplementedError("only the argument set of hifigan.py:163-178 is built")
    return _ops.mel_spectrogram(audio.to(device), sample_rate, hop_length, win_length, n_mels, f_min, f_max), None


def align_waveforms(ref_wav, deg_wav):
    """hifigan.py:113-136: shift of the cross-correlation peak, then trim to the common length.  Host-side
    utility of the band-swap data generator: FFT cross-correlation through torch
    instead of the reference's O(N*M) ``F.conv1d``; same argmax, same slicing."""
    ref_wav = ref_wav.view(1, 1, -1)
    deg_wav = deg_wav.view(1, 1, -1)
    with torch.no_grad():
        padding = deg_wav.shape[-1]
        n = ref_wav.shape[-1] + 2 * padding
        nfft = 1 << (n + padding - 1).bit_length()
        a = torch.fft.rfft(F.pad(ref_wav, (padding, padding)).double(), nfft)
        b = torch.fft.rfft(deg_wav.double(), nfft)
        cc = torch.fft.irfft(a * b.conj(), nfft)[..., : n - padding + 1]
        shift = int(torch.argmax(cc).item()) - padding
        if shift > 0:
            ref_aligned = ref_wav[..., shift:]
            deg_aligned = deg_wav[..., : ref_aligned.shape[-1]]
        else:
            deg_aligned = deg_wav[..., -shift:]
            ref_aligned = ref_wav[..., : deg_aligned.shape[-1]]
        min_len = min(ref_aligned.shape[-1], deg_aligned.shape[-1])
        return ref_aligned[..., :min_len], deg_aligned[..., :min_len]
