"""Drop-in for the reference's ``hifigan`` module (hifigan.py:106-136, 163-180): ``hifi_gan.decode_batch``,
``mel_spectogram``, ``align_waveforms`` and the band-swap data generator (hifigan.py:139-230, SURVEY.md §8(f)
rank 2) as functions.  Importing it runs nothing (the reference runs its dataset loop at import) and fetches nothing: the generator weights
come from ``ADDVISOR_HIFIGAN`` (a ``.pth`` state dict named like ``addvisor_hip.synthetic.hifigan_weights``,
weight-norm folded) or the seeded synthetic generator.  ``ADDVISOR_HIFIGAN_PADDING`` = zeros (default) | reflect and
``ADDVISOR_HIFIGAN_INFERENCE_PADDING`` = 0 (default) | n select the SpeechBrain wrapper's conv padding mode and mel edge
padding (addvisor_hip/hifigan.py: unverifiable offline, so they are options; a real SpeechBrain checkpoint most likely
wants reflect / 5)."""
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
            self._net = HipHifigan(cfg, sd, _rt.device(), padding_mode=os.environ.get("ADDVISOR_HIFIGAN_PADDING", "zeros"),
                                   inference_padding=int(os.environ.get("ADDVISOR_HIFIGAN_INFERENCE_PADDING", "0")))
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


# --------------------------------------------------------------------------- band-swap data generator (hifigan.py:139-230)
N_BANDS, BAND_BINS = 8, 64        # linspace(0, 8000, 513): 1 kHz = 64 bins; bin 512 (8 kHz) is in no band


def band_swap_variants(signal, waveform_voc=None):
    """One file of the reference loop (hifigan.py:159-228): vocode ``signal [L]`` (mel -> HiFi-GAN), align, Hann-1024 /
    hop-256 STFT of both, and for every 1 kHz band swap the vocoded complex bins into the original spectrogram and
    invert.  Returns ``(waves [8, L'], leakage [8])``; ``L' = 256 * (T - 1)`` as ``torch.istft`` without ``length``.
    All transforms are the HIP kernels (``advh_stft_forward``, ``advh_istft_bandswap``: one launch for the 8 bands).
    ``leakage`` (hifigan.py:214-219, the energy change outside the swapped band) is zero by construction here: the
    kernel selects whole bins from one spectrogram or the other and never writes a combined spectrogram."""
    signal = signal.reshape(-1).to(device, torch.float32)
    if waveform_voc is None:
        spectrogram, _ = mel_spectogram(audio=signal, sample_rate=16000, hop_length=256, win_length=1024, n_mels=80,
                                        n_fft=1024, f_min=0.0, f_max=8000.0, power=1, normalized=False,
                                        min_max_energy_norm=True, norm="slaney", mel_scale="slaney", compression=True)
        waveform_voc = hifi_gan.decode_batch(spectrogram.unsqueeze(0))
    sig_aligned, voc_aligned = align_waveforms(signal.view(1, 1, -1), waveform_voc.to(device))
    s_ref, s_voc = sig_aligned.reshape(1, -1).contiguous(), voc_aligned.reshape(1, -1).contiguous()
    n = s_ref.shape[-1]
    window = torch.hann_window(1024, device=device)
    X_r, _, _ = _ops.stft_forward(s_ref, n, 256, 1024, window=window, want_mag=False, want_phase=False)
    X_v, _, _ = _ops.stft_forward(s_voc, n, 256, 1024, window=window, want_mag=False, want_phase=False)
    T = X_r.shape[-1]
    waves = _ops.istft_bandswap(X_r, X_v, 256 * (T - 1), 0, BAND_BINS, N_BANDS, hop=256, win=1024, window=window)[:, 0]
    leakage = torch.zeros(N_BANDS)
    return waves, leakage


def generate_band_swap_dataset(wav_dir, output_dir, file_names=None, metadata_path=None, limit=5000, progress=False):
    """hifigan.py:139-230 as a function: ``<name>_vocoded_<start>-<end>.wav`` (float32 WAV, 16 kHz) for the eight bands
    of every file.  ``file_names`` defaults to the first CSV field of each metadata line, else the directory listing."""
    from addvisor_hip.wavio import read_wav, write_wav
    os.makedirs(output_dir, exist_ok=True)
    if file_names is None:
        if metadata_path and os.path.exists(metadata_path):
            with open(metadata_path, "r") as f:
                file_names = [line.strip().split(",")[0] for line in f]
        else:
            file_names = sorted(f for f in os.listdir(wav_dir) if f.endswith(".wav"))
    file_names = file_names[:limit]
    written = []
    for file_name in file_names:
        full_path = os.path.join(wav_dir, file_name)
        if not os.path.exists(full_path):
            continue
        audio, sr = read_wav(full_path)
        audio = audio.mean(0)
        if sr != 16000:                    # librosa.load(path, sr=16000) resamples (hifigan.py:156)
            import numpy as np
            from scipy.signal import resample_poly
            g = np.gcd(int(sr), 16000)
            audio = torch.from_numpy(resample_poly(audio.numpy(), 16000 // g, sr // g).astype(np.float32))
        waves, _ = band_swap_variants(audio)
        for z in range(N_BANDS):
            start = 1000 * z
            out_name = f"{file_name}_vocoded_{start}-{start + 1000}.wav"
            write_wav(os.path.join(output_dir, out_name), waves[z].unsqueeze(0).cpu(), 16000)
            written.append(out_name)
        if progress:
            print(f"{file_name}: {N_BANDS} band-swapped files")
    return written
