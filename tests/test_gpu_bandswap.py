"""GPU parity of the band-swap data generator (hifigan.py:139-230,
train_logReg_swapping.py:57-92) against the CPU oracle.  fp32 FFTs: |err| <= 2e-4 on unit-scale waveforms;
embedder features (fp16 GEMM operands): the tolerance of tests/test_gpu_embedder.py."""
import os

import numpy as np
import pytest
import torch

from addvisor_hip import ops, runtime, synthetic as syn
from addvisor_hip.wavio import read_wav, write_wav
from oracle import hifigan_ref, signal_ref, wav2vec2_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture
def tiny_runtime():
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    runtime.reset()
    yield
    os.environ.pop("ADDVISOR_EMBEDDER", None)
    runtime.reset()


def fake_vocoded(ref, shift, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    voc = torch.roll(ref, -shift)[: ref.numel() - 200] * 0.9
    return voc + 0.05 * torch.from_numpy(r.standard_normal(voc.numel()).astype(np.float32))


@pytest.mark.parametrize("n,shift", [(20000, 37), (33333, -120), (16000, 0)])
def test_band_swap_hann_matches_oracle(gpu_device, n, shift):
    import hifigan
    ref = syn.make_clips(1, n, seed=n)[0]
    voc = fake_vocoded(ref, shift, n + 1)
    waves, leak = hifigan.band_swap_variants(ref, waveform_voc=voc.view(1, 1, -1))
    r_al, v_al = hifigan_ref.align_waveforms(ref, voc)
    ref_waves, ref_leak = hifigan_ref.band_swap_hann(r_al.reshape(-1), v_al.reshape(-1))
    assert waves.shape == ref_waves.shape and waves.shape[0] == 8
    err = (waves.cpu() - ref_waves).abs().max().item()
    print(f"band swap (Hann 1024/256) n={n}: max err {err:.2e}")
    assert err < 2e-4 and ref_leak.max().item() < 1e-6 and leak.max().item() == 0.0
    # size-independent property: the eight band-swapped signals minus the original add up to (vocoded - original)
    # restricted to bins 0..511, i.e. swapping every band at once = one ISTFT of the fully swapped spectrogram
    window = torch.hann_window(1024)
    kw = dict(n_fft=1024, hop_length=256, win_length=1024, window=window)
    X_r = torch.stft(r_al.reshape(-1), return_complex=True, **kw)
    X_v = torch.stft(v_al.reshape(-1), return_complex=True, **kw)
    X_all = X_v.clone(); X_all[512] = X_r[512]
    base, full = torch.istft(X_r, **kw), torch.istft(X_all, **kw)
    assert ((waves.cpu() - base).sum(0) - (full - base)).abs().max().item() < 2e-3


def test_istft_bandswap_argument_errors(gpu_device):
    X = torch.zeros(1, 513, 63, dtype=torch.complex64, device=gpu_device)
    with pytest.raises(ValueError):
        ops.istft_bandswap(X.real, X, 62 * 256, hop=256, win=1024)
    with pytest.raises(RuntimeError):
        ops.istft_bandswap(X, X, 62 * 256, k0=0, kw=65, nbands=8, hop=256, win=1024)        # 8 * 65 > 513 bins


def test_band_swap_features_match_oracle(gpu_device, tiny_runtime):
    import train_logReg_swapping as T
    w = syn.make_clips(2, 80000, seed=91)
    feats = T.band_swap_features(w[0], w[1])
    cfg, sd = runtime.embedder_config_and_weights()
    fakes = signal_ref.band_swap_rect(w[0], w[1], audio_length=5)
    ref = wav2vec2_ref.extract_features(torch.cat([w[0:1], fakes], 0), sd, cfg).mean(dim=1)
    assert feats.shape == ref.shape == (9, cfg.hidden_size)
    err = (feats.cpu() - ref).abs().max().item()
    print("band-swap feature max err", err, "ref absmax", ref.abs().max().item())
    assert err < 2e-2


def test_dataset_generators_end_to_end(gpu_device, tiny_runtime, tmp_path):
    """hifigan.py:139-230 then train_logReg_swapping.py:30-128 on a synthetic three-file corpus."""
    import hifigan
    import train_logReg_swapping as T
    wav_dir, voc_dir, swap_dir = tmp_path / "wavs", tmp_path / "voc", tmp_path / "swapped"
    wav_dir.mkdir(); voc_dir.mkdir()
    names = []
    for i, n in enumerate((12000, 16000, 20480)):
        w = syn.make_clips(1, n, seed=200 + i)[0]
        write_wav(wav_dir / f"f{i}.wav", w, 16000, encoding="pcm16")
        write_wav(voc_dir / f"f{i}.wav", fake_vocoded(w, 5, 300 + i), 16000)
        names.append(f"f{i}.wav")
    (tmp_path / "meta.txt").write_text("".join(f"{n},spoof\n" for n in names))
    written = hifigan.generate_band_swap_dataset(str(wav_dir), str(swap_dir), metadata_path=str(tmp_path / "meta.txt"))
    assert len(written) == 24 and "f0.wav_vocoded_3000-4000.wav" in written
    a, sr = read_wav(swap_dir / written[0])
    assert sr == 16000 and a.shape[0] == 1 and a.shape[1] % 256 == 0 and torch.isfinite(a).all()
    X, y = T.generate_time_swap_dataset(str(tmp_path / "meta.txt"), save_dir=str(tmp_path / "feat"), dir_real=str(wav_dir),
                                        dir_vocoded=str(voc_dir))
    assert X.shape == (27, runtime.embedder_config_and_weights()[0].hidden_size) and y.tolist() == ([0] + [1] * 8) * 3
    assert np.isfinite(X).all() and os.path.exists(tmp_path / "feat" / "X_vocoded_anyband_16k.npy")
