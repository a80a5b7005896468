"""GPU parity of the HiFi-GAN V1 generator (HIP) against the CPU oracle (oracle/hifigan_ref.py; parity
unpinned: SpeechBrain is absent, the oracle restates the published V1 generator).

Stated tolerances on a waveform in [-1, 1] after ~50 stacked convolutions: the fp32-class mode (the default; the reference runs
the vocoder in fp32) max |err| <= 1e-4, mean <= 1e-5; the fp16-operand mode (precision="f16") max <= 2e-2, mean <= 2e-3."""
import numpy as np
import pytest
import torch

from addvisor_hip import ops, synthetic as syn
from addvisor_hip.hifigan import HipHifigan
from oracle import hifigan_ref, signal_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL = {"f32": (1e-4, 1e-5), "f16": (2e-2, 2e-3)}
TOL_MAX, TOL_MEAN = TOL["f16"]


def run(cfg, B, T, dev, seed, precision):
    sd = syn.hifigan_weights(cfg)
    r = np.random.Generator(np.random.PCG64(seed))
    mel = torch.from_numpy(r.normal(-4.0, 2.0, size=(B, cfg.in_channels, T)).astype(np.float32))
    net = HipHifigan(cfg, sd, dev, precision=precision)
    assert net.precision == precision
    wav = net.decode_batch(mel.to(dev))
    ref = hifigan_ref.generator(mel, sd, cfg)
    assert wav.shape == ref.shape == (B, 1, T * cfg.hop)
    err = (wav.cpu() - ref).abs()
    print(f"hifigan [{precision}] B={B} T={T}: max err {err.max():.3e} mean {err.mean():.3e} ref absmax {ref.abs().max():.3f}")
    assert err.max().item() <= TOL[precision][0] and err.mean().item() <= TOL[precision][1]
    return net, mel, wav


def test_default_precision_is_the_paths(gpu_device, monkeypatch):
    """HipHifigan() follows ADDVISOR_PRECISION like every other model class (default f32)."""
    cfg = syn.hifigan_tiny_config()
    sd = syn.hifigan_weights(cfg)
    monkeypatch.delenv("ADDVISOR_PRECISION", raising=False)
    assert HipHifigan(cfg, sd, gpu_device).precision == "f32"
    monkeypatch.setenv("ADDVISOR_PRECISION", "f16")
    net = HipHifigan(cfg, sd, gpu_device)
    assert net.precision == "f16" and net.with_precision("f16") is net and net.with_precision("f32").precision == "f32"


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_tiny_generator(gpu_device, precision):
    run(syn.hifigan_tiny_config(), 3, 12, gpu_device, 1, precision)
    run(syn.hifigan_tiny_config(), 1, 5, gpu_device, 2, precision)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_v1_generator(gpu_device, precision):
    net, mel, wav = run(syn.HifiganConfig(), 2, 24, gpu_device, 3, precision)
    one = net.decode_batch(mel[:1].to(gpu_device))
    assert torch.equal(one[0], wav[0])                                   # batch invariance, bit-exact


def test_mel_front_end(gpu_device):
    """hifigan.py:163-178: Hann-1024 / hop-256 STFT magnitude -> slaney mel -> log(clamp)."""
    w = syn.make_clips(2, 16000, seed=4)
    ref = signal_ref.mel_spectrogram(w)
    mel = ops.mel_spectrogram(w.to(gpu_device))
    assert mel.shape == ref.shape == (2, 80, 63)
    assert (mel.cpu() - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("padding_mode,inference_padding", [("reflect", 0), ("zeros", 5), ("reflect", 5)])
def test_speechbrain_wrapper_options(gpu_device, padding_mode, inference_padding):
    """The two wrapper choices that cannot be verified offline (hifigan.py:106-110: SpeechBrain's Conv1d defaults to
    padding_mode="reflect"; its generator's inference() replicates `inference_padding` mel frames on both sides) as options
    of the HIP generator, each against the oracle run with the same option; zeros / 0 stay the defaults."""
    cfg = syn.HifiganConfig()
    sd = syn.hifigan_weights(cfg)
    r = np.random.Generator(np.random.PCG64(5))
    mel = torch.from_numpy(r.normal(-4.0, 2.0, size=(2, cfg.in_channels, 40)).astype(np.float32))
    net = HipHifigan(cfg, sd, gpu_device, padding_mode=padding_mode, inference_padding=inference_padding, precision="f16")
    wav = net.decode_batch(mel.to(gpu_device))
    ref = hifigan_ref.generator(mel, sd, cfg, padding_mode=padding_mode, inference_padding=inference_padding)
    assert wav.shape == ref.shape == (2, 1, (40 + 2 * inference_padding) * cfg.hop)
    err = (wav.cpu() - ref).abs()
    plain = hifigan_ref.generator(mel, sd, cfg)
    edge = (ref[..., : plain.shape[-1]] - plain).abs().max().item() if inference_padding == 0 else float("nan")
    print(f"hifigan {padding_mode} / inference_padding {inference_padding}: max err {err.max():.3e} mean {err.mean():.3e}; "
          f"option vs published model at the edges: {edge:.3e}")
    assert err.max().item() <= TOL_MAX and err.mean().item() <= TOL_MEAN
    again = net.decode_batch(mel.to(gpu_device))                          # halos are refilled on every call
    assert torch.equal(again, wav)


@pytest.mark.parametrize("padding_mode,inference_padding", [("zeros", 0), ("reflect", 5)])
def test_v1_generator_f32_mode(gpu_device, padding_mode, inference_padding):
    """The fp32-class mode of the explanation path (split-format maps, three MFMAs per product) on the vocoder: ~50 stacked
    convolutions to 2e-5 of the oracle (fp32 CPU; stated tolerance 1e-4 on a waveform in [-1, 1]) where the fp16 mode gives 1.4e-3."""
    cfg = syn.HifiganConfig()
    sd = syn.hifigan_weights(cfg)
    r = np.random.Generator(np.random.PCG64(6))
    mel = torch.from_numpy(r.normal(-4.0, 2.0, size=(2, cfg.in_channels, 24)).astype(np.float32))
    net = HipHifigan(cfg, sd, gpu_device, padding_mode=padding_mode, inference_padding=inference_padding, precision="f32")
    wav = net.decode_batch(mel.to(gpu_device))
    ref = hifigan_ref.generator(mel, sd, cfg, padding_mode=padding_mode, inference_padding=inference_padding)
    assert wav.shape == ref.shape
    err = (wav.cpu() - ref).abs()
    print(f"hifigan f32 mode ({padding_mode}, pad {inference_padding}): max err {err.max():.3e} mean {err.mean():.3e}")
    assert err.max().item() <= 1e-4 and err.mean().item() <= 1e-5
    assert torch.equal(net.decode_batch(mel[:1].to(gpu_device))[0], wav[0])     # batch invariance


def test_v1_f32_fused_resblock_matches_unfused(gpu_device):
    """csrc/resblock_pair_x3.hip (the 32-channel stage's ResBlock steps as ONE fused split-format kernel each: the north-star's "MRF
    dilated-ResBlock Conv1d as LDS line-tile kernels" in the fp32-class mode) against the layer-by-layer x3 implicit-GEMM path: same
    three-MFMA arithmetic in the same K order, so the waveforms agree to the last bits of the split representation (stated 2e-6
    on a waveform in [-1, 1]); clip boundaries inside a tile, a ragged last tile and batch invariance included."""
    cfg = syn.HifiganConfig()
    sd = syn.hifigan_weights(cfg)
    r = np.random.Generator(np.random.PCG64(8))
    mel = torch.from_numpy(r.normal(-4.0, 2.0, size=(3, cfg.in_channels, 37)).astype(np.float32))
    fused = HipHifigan(cfg, sd, gpu_device, precision="f32")
    plain = HipHifigan(cfg, sd, gpu_device, precision="f32", fuse=False)
    kinds = [type(s[1]).__name__ for s in fused._workspace(3, 37)["steps"] if s[0] == "gemm"]
    assert kinds.count("ResblockPairX3Plan") == 9 and not any("X3" in type(s[1]).__name__ for s in plain._workspace(3, 37)["steps"] if s[0] == "gemm")
    a, b = fused.decode_batch(mel.to(gpu_device)), plain.decode_batch(mel.to(gpu_device))
    err = (a - b).abs().max().item()
    ref = hifigan_ref.generator(mel, sd, cfg)
    print(f"f32 fused vs unfused ResBlock steps: {err:.3e}; fused vs oracle {(a.cpu() - ref).abs().max():.3e}, unfused vs oracle {(b.cpu() - ref).abs().max():.3e}")
    assert err <= 2e-6
    assert (a.cpu() - ref).abs().max().item() <= 1e-4
    assert torch.equal(fused.decode_batch(mel[1:2].to(gpu_device))[0], a[1])          # batch invariance


@pytest.mark.parametrize("k,dil,B,T", [(7, 3, 3, 700), (11, 5, 2, 1000), (11, 1, 1, 255), (7, 1, 5, 64), (3, 5, 2, 300)])
def test_split_line_tile_matches_implicit_gemm(gpu_device, k, dil, B, T):
    """``advh_conv_taps_split`` (64 channels, weights streamed tap by tap through an LDS ring, eight wavefronts) against the x3 implicit
    GEMM on the same split-format maps: both ResBlock roles -- conv1 (bias + LeakyReLU) and conv2 (bias + residual, raw and pre-activated
    outputs) -- with clips shorter and longer than a 256-position tile; same three-MFMA arithmetic in the same K order."""
    from addvisor_hip import gemm as G
    g = torch.Generator().manual_seed(100 * k + dil)
    halo = 32
    maps = [G.Map1D(B, T, 64, halo, split=True).alloc(gpu_device) for _ in range(8)]
    src, res, o1, o2, p1, p2, q1, q2 = maps
    for m_ in (src, res):
        m_.t[:, :, halo:halo + T] = G.split_planes(torch.randn(B, T, 64, generator=g)).to(gpu_device)
    w = torch.randn(64, 64, k, generator=g) * (0.3 / k ** 0.5)
    b = torch.randn(64, generator=g) * 0.1
    assert G.taps_split_supported(src, o1, w, dil, min_k=3)
    # conv1 role
    G.plan_conv1d_taps(src, o1, w, b, dilation=dil, act="leaky", slope=0.1, device=gpu_device).run(src.t, out_h=o1.t)
    G.plan_conv1d_same(src, o2, w, b, dilation=dil, act="leaky", slope=0.1, device=gpu_device).run(src.t, out_h=o2.t)
    # conv2 role
    G.plan_conv1d_taps(src, p1, w, b, dilation=dil, slope2=0.1, device=gpu_device).run(src.t, out_h=p1.t, resid=res.t, out_h2=q1.t)
    G.plan_conv1d_same(src, p2, w, b, dilation=dil, slope2=0.1, device=gpu_device).run(src.t, out_h=p2.t, resid=res.t, out_h2=q2.t)
    torch.cuda.synchronize()
    for name, a_, b_ in (("conv1", o1, o2), ("conv2", p1, p2), ("conv2 pre-activated copy", q1, q2)):
        ja, jb = G.join_planes(a_.t.cpu()), G.join_planes(b_.t.cpu())
        err = float((ja - jb).abs().max() / jb.abs().max())
        print(f"split line tile vs implicit GEMM, k={k} d={dil} {name}: {err:.2e}")
        assert err <= 1e-6, name
        assert float(ja[:, :halo].abs().max()) == 0.0 and float(ja[:, halo + T:].abs().max()) == 0.0      # the halo stays zero
