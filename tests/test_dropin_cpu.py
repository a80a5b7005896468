"""CPU-only: host logic of the drop-in modules and of the utterance sharding (gloo, world_size 2)."""
import inspect
import os
import struct
import wave

import numpy as np
import pytest
import torch

from addvisor_hip import pipeline as P, synthetic as syn


def test_modules_import_without_gpu_or_network():
    from addvisor_hip import runtime
    os.environ.pop("ADDVISOR_EMBEDDER", None)
    runtime.reset()
    import addvisor
    import audioprocessor
    import captum_saliency
    import classifier_embedder
    import LMAC_metrics
    import loss_function
    assert hasattr(captum_saliency, "Wav2vec2LogReg") and hasattr(captum_saliency, "compute_camptum_saliency_metrics")
    assert addvisor.ADDvisor is addvisor.UNet
    sig = inspect.signature(audioprocessor.AudioProcessor.__init__)
    assert [p for p in sig.parameters][1:] == ["sampling_rate", "n_fft", "hop_length", "win_length", "n_mels", "audio_length"]
    d = {k: v.default for k, v in sig.parameters.items() if k != "self"}
    assert d == dict(sampling_rate=16000, n_fft=1024, hop_length=322, win_length=644, n_mels=80, audio_length=5)
    for name in ("compute_fidelity", "get_score_for_predicted_class", "compute_faithfulness", "compute_AD",
                 "compute_AI", "compute_AG", "extract_wavs", "AudioDataset", "collate_fn", "run_addvisor_metrics"):
        assert hasattr(LMAC_metrics, name)
    lm = loss_function.LMACLoss()
    assert lm.w_raw.tolist() == [3.0, 0.5, 3.0] and torch.allclose(lm.w, torch.nn.functional.softplus(lm.w_raw))
    clf = classifier_embedder.classifier
    assert clf.coef_.shape == (1, 768) and clf.intercept_.shape == (1,)
    import hifigan
    import train_addvisor
    import train_logReg_swapping
    for name in ("extract_wavs", "AudioDataset", "collate_fn", "train_addvisor", "UNet", "LMACLoss"):
        assert hasattr(train_addvisor, name)                        # importing runs nothing (the reference trains at import)
    assert hasattr(hifigan, "generate_band_swap_dataset") and hasattr(train_logReg_swapping, "train_logReg_timeswap")


def test_unet_state_dict_layout():
    import addvisor
    u = addvisor.UNet()
    ref = syn.unet_weights()
    assert set(u.state_dict().keys()) == set(ref.keys())
    for k, v in u.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    u.load_state_dict({"module." + k: v for k, v in ref.items()})       # DDP prefix (LMAC_metrics.py:23-25)
    assert torch.equal(u.state_dict()["e1.block.0.weight"], ref["e1.block.0.weight"])


def test_load_audio_and_metadata(tmp_path):
    import audioprocessor
    import LMAC_metrics
    x = (np.sin(np.arange(8000) / 10.0) * 20000).astype("<i2")
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(16000)
        w.writeframes(x.tobytes())
    ap = audioprocessor.AudioProcessor(audio_length=1)
    a, sr = ap.load_audio(str(p))
    assert sr == 16000 and a.shape == (16000,) and a.dtype == torch.float32
    assert torch.allclose(a[:8000], torch.from_numpy(x.astype(np.float32) / 32768.0)) and (a[8000:] == 0).all()
    a2, _ = audioprocessor.AudioProcessor(audio_length=0.25).load_audio(str(p))
    assert a2.shape == (4000,)
    meta = tmp_path / "meta.txt"
    meta.write_text("x/a.wav,1,foo\ny/b.wav,0\n")
    assert LMAC_metrics.extract_wavs(str(meta)) == ["x/a.wav", "y/b.wav"]


def test_stft_argument_errors():
    import audioprocessor
    ap = audioprocessor.AudioProcessor(audio_length=1)
    with pytest.raises(ValueError, match="waveform must be 1D"):
        ap.compute_stft(torch.zeros(2, 3, 16000))
    with pytest.raises(ValueError, match="ISTFT expects complex input!"):
        ap.compute_invert_stft(torch.zeros(1, 513, 50))


def test_shard_indices_cover_and_are_disjoint():
    for n, w in ((71237, 8), (10, 4), (3, 8), (64, 1)):
        seen = []
        for r in range(w):
            seen += list(P.shard_indices(n, r, w))
        assert seen == list(range(n))


def _worker(rank, world, n_total, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).uniform(0, 1, size=(n_total, 3)).astype(np.float32))
    idx = P.shard_indices(n_total, rank, world)
    got = P.gather_probabilities(full[idx.start:idx.stop].clone(), n_total)
    q.put((rank, torch.equal(got, full)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [37, 64])
def test_gather_probabilities_gloo_world2(n_total):
    """The one exchange step of the sharded run: every rank ends with the same [N,3] table in clip order,
    so the metric reduction is bit-identical to the 1-rank run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total
    procs = [ctx.Process(target=_worker, args=(r, 2, n_total, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    assert all(ok for _, ok in res) and all(p.exitcode == 0 for p in procs)


def test_hifigan_module_and_align():
    import hifigan
    from oracle import hifigan_ref
    assert hasattr(hifigan.hifi_gan, "decode_batch")
    r = np.random.Generator(np.random.PCG64(9))
    ref = torch.from_numpy(r.standard_normal(3000).astype(np.float32))
    for shift in (37, -52, 0):
        deg = torch.roll(ref, -shift)[:2800] + 0.01 * torch.from_numpy(r.standard_normal(2800).astype(np.float32))
        a1, b1 = hifigan.align_waveforms(ref, deg)
        a2, b2 = hifigan_ref.align_waveforms(ref, deg)
        assert a1.shape == a2.shape == (1, 1, a2.shape[-1]) and torch.equal(a1, a2) and torch.equal(b1, b2)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import addvisor
    from oracle import unet_ref
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                           # same initial weights on every rank
    net = addvisor.UNet()                                          # the drop-in module = the parameter container DDP sees
    net.train()

    class OracleForward(torch.nn.Module):
        """The product has no CPU arithmetic (UNet.forward raises off the GPU), so this CPU rehearsal of the data-parallel
        step runs the ORACLE's forward over the drop-in module's parameters: what is under test is that those
        parameters are ordinary nn.Parameters whose gradients DistributedDataParallel averages."""

        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            sd = dict(self.m.named_parameters())
            sd.update(dict(self.m.named_buffers()))
            return unet_ref.unet_forward(x, sd, bn_batch=True)

    ddp = DDP(OracleForward(net))
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(100 + rank)).uniform(0, 2, size=(1, 1, 32, 8)).astype(np.float32))
    mask = ddp(x)
    try:                                                           # and the product really refuses to compute off the GPU
        with torch.enable_grad():
            net(x)
        ok_fwd = False
    except RuntimeError:
        ok_fwd = bool(torch.isfinite(mask).all())
    (mask * (rank + 1)).mean().backward()                         # different local losses; DDP averages the gradients
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, ok_fwd and all(torch.equal(g, gathered[0]) for g in gathered) and bool(flat.abs().sum() > 0)))
    dist.destroy_process_group()


def test_unet_training_forward_under_ddp_gloo_world2():
    """Training step, data-parallel (train_addvisor.py:410-412 hands the model to accelerate = DDP): the drop-in UNet's
    parameters are ordinary nn.Parameters, so the gradient all-reduce (RCCL on the GPUs, gloo here) needs no special
    casing.  The forward in this CPU rehearsal is the oracle's (the product computes on the GPU only and raises here)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in procs]
    [p.join(60) for p in procs]
    assert all(ok for _, ok in res) and all(p.exitcode == 0 for p in procs)


def test_wavio_round_trip_and_formats(tmp_path):
    from addvisor_hip.wavio import read_wav, write_wav
    r = np.random.Generator(np.random.PCG64(4))
    x = torch.from_numpy(r.uniform(-0.9, 0.9, size=(2, 1001)).astype(np.float32))
    write_wav(tmp_path / "f.wav", x, 22050)                                   # float32, stereo, odd frame count
    a, sr = read_wav(tmp_path / "f.wav")
    assert sr == 22050 and torch.equal(a, x)
    write_wav(tmp_path / "p.wav", x[0], 16000, encoding="pcm16")
    b, sr = read_wav(tmp_path / "p.wav")
    assert sr == 16000 and b.shape == (1, 1001) and (b[0] - x[0]).abs().max() <= 1 / 32768 + 1e-7
    with wave.open(str(tmp_path / "p.wav"), "rb") as w:                      # the stdlib reader agrees on the PCM file
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (16000, 1, 2, 1001)
    v = np.round(x[0].numpy() * 8388608.0).astype(np.int32)                   # hand-made 24-bit PCM
    raw = b"".join(struct.pack("<i", int(s))[:3] for s in v)
    hdr = struct.pack("<HHIIHH", 1, 1, 8000, 24000, 3, 24)
    (tmp_path / "t.wav").write_bytes(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack("<I", 16) + hdr
                                     + b"data" + struct.pack("<I", len(raw)) + raw + b"\x00")
    c, sr = read_wav(tmp_path / "t.wav")
    assert sr == 8000 and (c[0] - x[0]).abs().max() < 1e-6
    (tmp_path / "bad.wav").write_bytes(b"nope")
    with pytest.raises(ValueError):
        read_wav(tmp_path / "bad.wav")


def test_logreg_trainer_and_eer(tmp_path):
    """train_logReg_swapping.py:105-128 on separable synthetic features; the .joblib feeds runtime.classifier()."""
    import train_logReg_swapping as T
    from addvisor_hip import runtime
    r = np.random.Generator(np.random.PCG64(8))
    y = np.array(([0] + [1] * 8) * 40)
    X = r.standard_normal((y.size, 24)).astype(np.float32) + 2.5 * y[:, None] * np.linspace(-1, 1, 24)[None, :]
    model, acc, eer = T.train_logReg_timeswap(X, y, out_path=str(tmp_path / "ckpt" / "lr.joblib"))
    assert acc > 0.95 and 0.0 <= eer < 0.1
    s = model.predict_proba(X)[:, 1]
    assert abs(T.equal_error_rate(y, s) - T.equal_error_rate(y, s * 0.5)) < 1e-9       # EER is rank-based
    os.environ["ADDVISOR_LOGREG"] = str(tmp_path / "ckpt" / "lr.joblib")
    try:
        runtime.reset()
        clf = runtime.classifier()
        assert np.allclose(np.asarray(clf.coef_).reshape(-1), model.coef_.reshape(-1))
    finally:
        os.environ.pop("ADDVISOR_LOGREG", None)
        runtime.reset()
    (tmp_path / "m.txt").write_text("a.wav,x\nb.wav,y,z\n")
    assert T.find_all_files(str(tmp_path / "m.txt")) == ["a.wav", "b.wav"]


def test_bench_self_launch_starts_ranks_and_propagates_failure():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (round 1 silently measured one GPU).  Without a
    GPU every rank refuses to run ("bench.py needs a GPU"), which must surface as a non-zero exit of the parent -- and the
    message proves that two ranks with WORLD_SIZE = 2 were started and that the parent itself never touched the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--master-port", str(33000 + os.getpid() % 2000)], capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    assert out.count("bench.py needs a GPU") >= 1 and "--gpus 2 but WORLD_SIZE" not in out
    # under a launcher with the wrong world size it refuses
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in (r.stdout + r.stderr)


def test_collate_is_lazy_and_lazy_tensors_behave(monkeypatch):
    """SURVEY.md D12 / LMAC_metrics.py:109-114: ``collate_fn`` returns the reference's 5-tuple but does no device work of its
    own -- magnitude / phase / features are computed on first use only (the fused ``run_addvisor_metrics`` never uses them)."""
    import LMAC_metrics
    calls = {"stft": 0, "feat": 0}

    def fake_stft(w):
        calls["stft"] += 1
        return w * (1 + 0j), w.abs(), torch.zeros_like(w)

    def fake_feat(w):
        calls["feat"] += 1
        return w[:, None, :4].repeat(1, 3, 1)

    monkeypatch.setattr(LMAC_metrics.audio_processor, "compute_stft", fake_stft)
    monkeypatch.setattr(LMAC_metrics.audio_processor, "extract_features", fake_feat)
    clips = [(torch.full((8,), float(i) - 1.0), f"c{i}.wav") for i in range(3)]
    waves, mag, ph, feats, names = LMAC_metrics.collate_fn(clips)
    assert waves.shape == (3, 8) and names == ("c0.wav", "c1.wav", "c2.wav")
    assert calls == {"stft": 0, "feat": 0} and not mag.materialized and not feats.materialized
    assert isinstance(mag, LMAC_metrics.LazyTensor) and "pending" in repr(mag)
    # first use computes; magnitude and phase share ONE STFT; tensor protocol: attributes, indexing, arithmetic, torch.* calls
    assert mag.shape == (3, 8) and calls["stft"] == 1
    assert torch.equal(ph[0], torch.zeros(8)) and calls["stft"] == 1
    assert torch.equal(mag + 1, waves.abs() + 1) and torch.equal(2 * mag, 2 * waves.abs()) and torch.equal(torch.log1p(mag), torch.log1p(waves.abs()))
    assert torch.equal(feats.mean(dim=1), waves[:, :4]) and calls["feat"] == 1 and len(feats) == 3
    assert feats.materialized and mag.materialized


def test_loss_scaler_backs_off_and_regrows():
    from addvisor_hip.lmac_loss import LossScaler
    s = LossScaler(4096.0, growth_interval=3)
    assert s.backoff() and s.scale == 256.0                      # kept for the following steps
    for _ in range(3):
        s.good()
    assert s.scale == 512.0                                      # grows back by powers of two ...
    for _ in range(30):
        s.good()
    assert s.scale == 4096.0                                     # ... up to the initial value, never beyond
    tiny = LossScaler(2.0 ** -18)
    assert not tiny.backoff() and tiny.scale == 2.0 ** -18       # floor: the caller raises
