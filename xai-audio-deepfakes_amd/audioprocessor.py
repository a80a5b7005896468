"""Drop-in for the reference's ``audioprocessor`` module (audioprocessor.py:1-131): same class, method
names, argument meaning and errors; STFT / ISTFT / embedder run as hand-written gfx950 kernels."""
import numpy as np
import torch
import torch.nn.functional as F

from addvisor_hip import ops as _ops, runtime as _rt
from classifier_embedder import classifier, processor, wav2vec2, zero_mean_unit_var_norm  # noqa: F401


class _Accelerator:
    """Stand-in for ``accelerate.Accelerator()`` (audioprocessor.py:15): only ``.device`` is used."""

    @property
    def device(self):
        return _rt.device()


accelerator = _Accelerator()
device = accelerator.device
wav2vec2 = wav2vec2.to(device)
wav2vec2.eval()


def _read_wav(path):
    """WAV reader (the reference uses torchaudio.load, audioprocessor.py:50): ``[channels, frames]`` fp32, rate."""
    from addvisor_hip.wavio import read_wav
    return read_wav(path)


class AudioProcessor:
    def __init__(self, sampling_rate=16000, n_fft=1024, hop_length=322, win_length=644, n_mels=80, audio_length=5):
        if n_fft != 1024:
            raise ValueError("the HIP STFT kernels are built for n_fft=1024 (the reference's only size)")
        self.sampling_rate = sampling_rate
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.n_mels = n_mels
        self.audio_length = audio_length

    # audioprocessor.py:49-63
    def load_audio(self, audio_path, target_sr=16000):
        audio, sr = _read_wav(audio_path)
        if audio.ndim > 1:
            audio = audio.squeeze(0)
        if sr != target_sr:
            from scipy.signal import resample_poly
            g = np.gcd(int(sr), int(target_sr))
            audio = torch.from_numpy(resample_poly(audio.numpy(), target_sr // g, sr // g).astype(np.float32))
        length = int(self.audio_length * target_sr)
        current_length = audio.shape[0]
        if current_length < length:
            audio = F.pad(audio, (0, length - current_length))
        else:
            audio = audio[:length]
        return audio, target_sr

    # audioprocessor.py:69-77
    def extract_features(self, waveforms):
        if waveforms.dim() == 1:
            waveforms = waveforms[None]
        x = waveforms.to(device, torch.float32)
        hid, _, _ = _rt.hip_embedder().forward(x)          # normaliser + wav2vec2 -> hidden_states[9]
        return hid.squeeze(0)

    def classify(self, waveforms):
        """Pool + logreg head fused behind the embedder (LMAC_metrics.py:130): ``(logits, probs) [B,1]``."""
        if waveforms.dim() == 1:
            waveforms = waveforms[None]
        _, logits, probs = _rt.hip_embedder().forward(waveforms.to(device, torch.float32), want_hidden=False)
        return logits, probs

    # audioprocessor.py:82-112
    def compute_stft(self, waveform):
        length = int(self.audio_length * self.sampling_rate)
        if waveform.dim() == 1:
            single = True
            waveform = waveform[None]
        elif waveform.dim() == 2:
            single = False
        else:
            raise ValueError("waveform must be 1D (single) or 2D (batched waveforms)")
        w = waveform.to(device, torch.float32)
        X, mag, phase = _ops.stft_forward(w, length, self.hop_length, self.win_length)   # pads / crops to `length`
        if single:
            return X[0], mag[0], phase[0]
        return X, mag, phase

    # audioprocessor.py:117-131
    def compute_invert_stft(self, spectrogram):
        if not torch.is_complex(spectrogram):
            raise ValueError("ISTFT expects complex input!")
        expected_length = int(self.audio_length * self.sampling_rate)
        single = spectrogram.dim() == 2
        s = spectrogram[None] if single else spectrogram
        out = _ops.istft_complex(s.to(device, torch.complex64), expected_length, self.hop_length, self.win_length)
        return out[0] if single else out
