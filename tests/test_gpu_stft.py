"""GPU parity: framed rFFT STFT and masked ISTFT kernels vs the CPU oracle and the golden vectors."""
import math

import numpy as np
import pytest
import torch

from addvisor_hip import ops, synthetic as syn
from oracle import signal_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# stated tolerances (fp32 transforms): spectra relative to the largest bin, waveforms absolute
TOL_SPEC = 2e-6      # |X - X_ref| <= TOL_SPEC * max|X_ref| + 1e-5
TOL_WAVE = 5e-6
TOL_PHASE = 2e-3     # radians, on bins with |X| > 1e-3 * max|X|, wrap-aware


def spec_close(a, b):
    a, b = a.cpu(), b.cpu()
    assert a.shape == b.shape
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= TOL_SPEC * scale + 1e-5, (err, scale)


def phase_close(ph, ph_ref, mag_ref):
    ph, ph_ref, mag_ref = ph.cpu(), ph_ref.cpu(), mag_ref.cpu()
    sel = mag_ref > 1e-3 * mag_ref.max()
    d = (ph - ph_ref)[sel]
    d = torch.remainder(d + math.pi, 2 * math.pi) - math.pi
    assert d.abs().max().item() <= TOL_PHASE


@pytest.mark.parametrize("sec,extra", [(4, 0), (4, -5000), (5, 777), (1, 0)])
def test_stft_vs_oracle(gpu_device, sec, extra):
    L = sec * 16000
    w = syn.make_clips(3, L + extra, seed=7)
    X, mag, ph = ops.stft_forward(w.to(gpu_device), L)
    Xr, magr, phr = signal_ref.compute_stft(w, audio_length=sec)
    spec_close(torch.view_as_real(X), torch.view_as_real(Xr))
    spec_close(mag, magr)
    phase_close(ph, phr, magr)
    # DC and Nyquist bins are exactly real, as in torch
    assert (X[:, 0].imag == 0).all() and (X[:, 512].imag == 0).all()


def test_stft_golden(gpu_device, golden):
    g = golden("stft_1s.npz")
    w = syn.make_clips(1, 16000, seed=21)
    X, mag, ph = ops.stft_forward(w.to(gpu_device), 16000)
    spec_close(X.real, torch.from_numpy(g["X_re"]))
    spec_close(X.imag, torch.from_numpy(g["X_im"]))
    spec_close(mag, torch.from_numpy(g["mag"]))
    phase_close(ph, torch.from_numpy(g["phase"]), torch.from_numpy(g["mag"]))
    for sec in (4, 5):
        g = golden(f"stft_{sec}s.npz")
        w = syn.make_clips(2, sec * 16000 + 777, seed=22)
        X, mag, ph = ops.stft_forward(w.to(gpu_device), sec * 16000)
        assert tuple(X.shape) == tuple(g["shape"])
        spec_close(mag[:, ::19, ::7], torch.from_numpy(g["mag"]))
        assert abs(mag.double().sum().item() - float(g["mag_sum"])) < 1e-5 * float(g["mag_sum"])


@pytest.mark.parametrize("sec", [4, 5])
def test_istft_roundtrip_and_golden(gpu_device, golden, sec):
    g = golden(f"stft_{sec}s.npz")
    L = sec * 16000
    w = syn.make_clips(2, L + 777, seed=22)
    X, mag, ph = ops.stft_forward(w.to(gpu_device), L)
    back = ops.istft_complex(X, L)
    assert (back.cpu() - w[:, :L]).abs().max().item() < TOL_WAVE          # STFT -> ISTFT round trip
    assert (back.cpu()[:, ::13] - torch.from_numpy(g["istft_roundtrip"])).abs().max().item() < TOL_WAVE
    m = torch.from_numpy(np.random.Generator(np.random.PCG64(23)).uniform(0, 1, size=tuple(mag.shape)).astype(np.float32))
    win, _ = ops.istft_masked(mag, ph, m.to(gpu_device), L, domain="linear", want_out=False)
    assert (win.cpu()[:, ::13] - torch.from_numpy(g["istft_masked"])).abs().max().item() < TOL_WAVE


@pytest.mark.parametrize("domain", ["linear", "log1p"])
def test_istft_masked_vs_oracle(gpu_device, domain):
    L = 64000
    w = syn.make_clips(3, L, seed=9)
    X, mag, ph = signal_ref.compute_stft(w, audio_length=4)
    r = np.random.Generator(np.random.PCG64(5))
    mask = torch.from_numpy(r.uniform(0, 1, size=(3, 512, 196)).astype(np.float32))
    rel, irr = signal_ref.apply_mask(signal_ref.embed_mask(mask, 513, 199), mag, ph, domain)
    ref_in = signal_ref.compute_invert_stft(rel, audio_length=4)
    ref_out = signal_ref.compute_invert_stft(irr, audio_length=4)
    d = gpu_device
    w_in, w_out = ops.istft_masked(mag.to(d), ph.to(d), mask.to(d), L, domain=domain)
    assert (w_in.cpu() - ref_in).abs().max().item() < TOL_WAVE
    assert (w_out.cpu() - ref_out).abs().max().item() < TOL_WAVE
    only_out = ops.istft_masked(mag.to(d), ph.to(d), mask.to(d), L, domain=domain, want_in=False)[1]
    assert torch.equal(only_out, w_out)
    if domain == "linear":                                               # linearity: in + out = identity
        assert ((w_in + w_out).cpu() - w).abs().max().item() < 2 * TOL_WAVE


@pytest.mark.parametrize("domain", ["linear", "log1p"])
def test_istft_masked_complex_vs_oracle(gpu_device, domain):
    """advh_istft_masked_c64 -- what the explanation pipeline runs: X' = X * g(mask, |X|) / |X| from the HIP forward's own
    complex spectrogram, against the reference formulation g e^{j angle X} (loss_function.py:36-45, LMAC_metrics.py:136-153)
    evaluated by the oracle; also against the (|X|, angle X) entry point, and the one-output calls against the two-output one."""
    L = 64000
    w = syn.make_clips(3, L, seed=9)
    _, mag, ph = signal_ref.compute_stft(w, audio_length=4)
    mask = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).uniform(0, 1, size=(3, 512, 196)).astype(np.float32))
    mask[0, :40, :30] = 0.0                                              # exact zeros and ones included
    mask[1, 100:140] = 1.0
    rel, irr = signal_ref.apply_mask(signal_ref.embed_mask(mask, 513, 199), mag, ph, domain)
    ref_in = signal_ref.compute_invert_stft(rel, audio_length=4)
    ref_out = signal_ref.compute_invert_stft(irr, audio_length=4)
    d = gpu_device
    X, gmag, gph = ops.stft_forward(w.to(d), L)
    w_in, w_out = ops.istft_masked_c64(X, mask.to(d), L, domain=domain)
    e_in, e_out = (w_in.cpu() - ref_in).abs().max().item(), (w_out.cpu() - ref_out).abs().max().item()
    p_in, p_out = ops.istft_masked(gmag, gph, mask.to(d), L, domain=domain)
    print(f"complex masked ISTFT ({domain}): err vs oracle {e_in:.2e} / {e_out:.2e}; vs polar entry point {(w_in - p_in).abs().max().item():.2e}")
    assert e_in < TOL_WAVE and e_out < TOL_WAVE
    assert (w_in - p_in).abs().max().item() < TOL_WAVE and (w_out - p_out).abs().max().item() < TOL_WAVE
    assert torch.equal(ops.istft_masked_c64(X, mask.to(d), L, domain=domain, want_in=False)[1], w_out)
    assert torch.equal(ops.istft_masked_c64(X, mask.to(d), L, domain=domain, want_out=False)[0], w_in)
    if domain == "linear":
        assert ((w_in + w_out).cpu() - w).abs().max().item() < 2 * TOL_WAVE


def test_istft_errors(gpu_device):
    with pytest.raises(ValueError, match="ISTFT expects complex input!"):
        ops.istft_complex(torch.zeros(1, 513, 199, device=gpu_device), 64000)
    with pytest.raises(ValueError):
        ops.stft_forward(torch.zeros(2, 3, 64000, device=gpu_device), 64000)


def test_stft_full_batch_property(gpu_device):
    """BASELINE size (B=64, 4 s): Parseval-type check, no oracle needed."""
    L = 64000
    w = syn.make_clips(64, L, seed=3).to(gpu_device)
    X, mag, _ = ops.stft_forward(w, L)
    back = ops.istft_complex(X, L)
    assert (back - w).abs().max().item() < TOL_WAVE
    assert torch.allclose(mag, X.abs(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("hop,win", [(322, 644), (321, 642), (256, 1024), (160, 400)])
def test_istft_overlap_add_without_atomics_roundtrip(gpu_device, hop, win):
    """Round 3: the inverse writes every frame's windowed samples into the frame's own LDS rows and sums the R overlapping frames at
    emit time (round 2 used LDS float atomics: 114 LDS cycles each).  STFT -> ISTFT is the identity for any hop / window length the
    kernel accepts: the reference's 644 / 322, a window with an ODD left offset (win % 4 == 2: scalar stores), Hann-size 1024 / 256
    (four overlapping frames, samples spill into the second row) and a short window."""
    L = 16000
    w = syn.make_clips(3, L, seed=55)
    X, _, _ = ops.stft_forward(w.to(gpu_device), L, hop, win)
    back = ops.istft_complex(X, L, hop, win)
    assert back.shape == (3, L)
    err = (back.cpu() - w).abs().max().item()
    print(f"roundtrip hop {hop} win {win}: {err:.2e}")
    assert err < TOL_WAVE
