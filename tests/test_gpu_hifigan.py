"""GPU parity of the HiFi-GAN V1 generator (HIP) against the CPU oracle (oracle/hifigan_ref.py; parity
unpinned: SpeechBrain is absent, the oracle restates the published V1 generator).

Stated tolerance: waveform in [-1, 1] after ~50 fp16 convolutions with fp32 accumulation:
max |err| <= 2e-2, mean |err| <= 2e-3."""
import numpy as np
import pytest
import torch

from addvisor_hip import ops, synthetic as syn
from addvisor_hip.hifigan import HipHifigan
from oracle import hifigan_ref, signal_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL_MAX, TOL_MEAN = 2e-2, 2e-3


def run(cfg, B, T, dev, seed):
    sd = syn.hifigan_weights(cfg)
    r = np.random.Generator(np.random.PCG64(seed))
    mel = torch.from_numpy(r.normal(-4.0, 2.0, size=(B, cfg.in_channels, T)).astype(np.float32))
    net = HipHifigan(cfg, sd, dev)
    wav = net.decode_batch(mel.to(dev))
    ref = hifigan_ref.generator(mel, sd, cfg)
    assert wav.shape == ref.shape == (B, 1, T * cfg.hop)
    err = (wav.cpu() - ref).abs()
    print(f"hifigan B={B} T={T}: max err {err.max():.3e} mean {err.mean():.3e} ref absmax {ref.abs().max():.3f}")
    assert err.max().item() <= TOL_MAX and err.mean().item() <= TOL_MEAN
    return net, mel, wav


def test_tiny_generator(gpu_device):
    run(syn.hifigan_tiny_config(), 3, 12, gpu_device, 1)
    run(syn.hifigan_tiny_config(), 1, 5, gpu_device, 2)


def test_v1_generator(gpu_device):
    net, mel, wav = run(syn.HifiganConfig(), 2, 24, gpu_device, 3)
    one = net.decode_batch(mel[:1].to(gpu_device))
    assert torch.equal(one[0], wav[0])                                   # batch invariance, bit-exact


def test_mel_front_end(gpu_device):
    """hifigan.py:163-178: Hann-1024 / hop-256 STFT magnitude -> slaney mel -> log(clamp)."""
    w = syn.make_clips(2, 16000, seed=4)
    ref = signal_ref.mel_spectrogram(w)
    mel = ops.mel_spectrogram(w.to(gpu_device))
    assert mel.shape == ref.shape == (2, 80, 63)
    assert (mel.cpu() - ref).abs().max().item() < 2e-3
