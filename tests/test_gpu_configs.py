"""GPU parity at the sizes BASELINE.json's configs 3, 4 and 5 name (the small-shape parity lives in the per-kernel test files).

  config 3  HiFi-GAN V1 on 251 mel frames (4 s): B = 2 against the oracle, B = 256 through size-independent properties
            (finite, |wav| <= 1, batch invariance bit-exact, run-to-run determinism); reference call: hifigan.py:180.
  config 4  run_addvisor_metrics (LMAC_metrics.py:117-172) over ragged batches of 4 s clips, wav2vec2-base.
  config 5  wav2vec2-LARGE (1024 / 16 heads / 4096, layer-norm feature extractor, pre-LN encoder) forward in both precisions
            against the oracle and the reference-generated fixture; IntegratedGradients n_steps = 50 (captum_saliency.py:131-135)
            on that model against the oracle, path-batched with a small internal batch.
"""
import os

import numpy as np
import pytest
import torch

from addvisor_hip import runtime, synthetic as syn
from addvisor_hip.embedder import HipEmbedder
from addvisor_hip.hifigan import HipHifigan
from oracle import attribution_ref, hifigan_ref, lmac_ref, wav2vec2_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


# ------------------------------------------------------------------------------------------ config 3
def _mel(B, T, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(r.normal(-4.0, 2.0, size=(B, 80, T)).astype(np.float32))


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_hifigan_v1_251_frames_vs_oracle(gpu_device, precision):
    """Stated tolerance on a waveform in [-1, 1]: fp32-class mode (default) max 1e-4, mean 1e-5; fp16 operands max 2e-2, mean 2e-3."""
    cfg = syn.HifiganConfig()
    sd = syn.hifigan_weights(cfg)
    mel = _mel(2, 251, 11)
    wav = HipHifigan(cfg, sd, gpu_device, precision=precision).decode_batch(mel.to(gpu_device))
    ref = hifigan_ref.generator(mel, sd, cfg)
    assert wav.shape == ref.shape == (2, 1, 251 * 256)
    err = (wav.cpu() - ref).abs()
    print(f"HiFi-GAN V1 [{precision}] 2 x 251 frames: max err {err.max():.3e} mean {err.mean():.3e} |ref|max {ref.abs().max():.3f}")
    tmax, tmean = (1e-4, 1e-5) if precision == "f32" else (2e-2, 2e-3)
    assert err.max().item() <= tmax and err.mean().item() <= tmean


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_hifigan_v1_batch256_properties(gpu_device, precision):
    cfg = syn.HifiganConfig()
    sd = syn.hifigan_weights(cfg)
    net = HipHifigan(cfg, sd, gpu_device, precision=precision)
    mel = _mel(256, 251, 12).to(gpu_device)
    wav = net.decode_batch(mel)
    assert wav.shape == (256, 1, 251 * 256)
    assert bool(torch.isfinite(wav).all()) and wav.abs().max().item() <= 1.0
    assert wav.std().item() > 1e-4                                          # not a constant
    again = net.decode_batch(mel)
    assert torch.equal(again, wav)                                          # deterministic
    pair = net.decode_batch(mel[[0, 255]].contiguous())                    # utterances are independent => sharding is exact
    assert torch.equal(pair[0], wav[0]) and torch.equal(pair[1], wav[255])


# ------------------------------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def large_model():
    cfg = syn.large_config()
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    return cfg, sd, coef, icpt


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_large_embedder_4s(gpu_device, golden, large_model, precision):
    """hidden_states[9] of a pre-LN encoder is the un-normalised residual stream, so the tolerance is relative to its
    magnitude: f32 mode 2e-5 * max|ref| (+1e-5), f16 mode 4e-3 * max|ref| (mean 4e-4 * max|ref|); logits 1e-4 / 1e-2."""
    cfg, sd, coef, icpt = large_model
    w = syn.make_clips(1, 64000)
    emb = HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision=precision)
    hid, logit, prob = emb.forward(w.to(gpu_device))
    ref_h = wav2vec2_ref.hidden_states(wav2vec2_ref.zero_mean_unit_var_norm(w), sd, cfg, upto=9)[9]
    ref_logit, _ = wav2vec2_ref.logreg(ref_h.mean(1), coef, icpt)
    amax = ref_h.abs().max().item()
    err = (hid.cpu() - ref_h).abs()
    le = (logit.cpu() - ref_logit).abs().max().item()
    print(f"wav2vec2-large {precision}: hidden max err {err.max():.3e} mean {err.mean():.3e}, |ref|max {amax:.2f}, logit err {le:.3e}")
    tol_max, tol_mean, tol_logit = (2e-5 * amax + 1e-5, 2e-6 * amax + 1e-6, 1e-4) if precision == "f32" else (4e-3 * amax, 4e-4 * amax, 1e-2)
    assert err.max().item() <= tol_max and err.mean().item() <= tol_mean and le <= tol_logit
    g = golden("embedder_large_4s.npz")                                    # the reference's own extract_features
    assert tuple(hid.shape[1:]) == tuple(g["shape"])
    assert (hid[0, :8, :16].cpu() - torch.from_numpy(g["corner"])).abs().max().item() <= tol_max
    assert (hid[0].mean(0).cpu() - torch.from_numpy(g["pooled"])).abs().max().item() <= tol_max


# ------------------------------------------------------------------------------------------ the reference's own embedder shape
@pytest.fixture(scope="module")
def xlsr2b_model():
    cfg = syn.xlsr2b_config(num_hidden_layers=10)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    return cfg, sd, coef, icpt


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_xlsr2b_embedder_4s(gpu_device, golden, xlsr2b_model, precision):
    """classifier_embedder.py:13-16, 25: XLS-R-2B at FULL width (hidden 1920, 16 heads x 120 -> the streaming head-dim-120
    attention, FFN 7680, layer-norm feature extractor, pre-LN encoder), truncated to the layers hidden_states[9] needs, one
    4 s clip (audioprocessor.py:69-77), against the oracle and the reference-generated fixture.  Tolerances as for the large
    model: relative to the un-normalised residual stream's magnitude."""
    cfg, sd, coef, icpt = xlsr2b_model
    w = syn.make_clips(1, 64000)
    emb = HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision=precision)
    hid, logit, prob = emb.forward(w.to(gpu_device))
    ref_h = wav2vec2_ref.hidden_states(wav2vec2_ref.zero_mean_unit_var_norm(w), sd, cfg, upto=9)[9]
    ref_logit, _ = wav2vec2_ref.logreg(ref_h.mean(1), coef, icpt)
    amax = ref_h.abs().max().item()
    err = (hid.cpu() - ref_h).abs()
    le = (logit.cpu() - ref_logit).abs().max().item()
    print(f"XLS-R-2B shape {precision}: hidden max err {err.max():.3e} mean {err.mean():.3e}, |ref|max {amax:.2f}, logit err {le:.3e}")
    tol_max, tol_mean, tol_logit = (2e-5 * amax + 1e-5, 2e-6 * amax + 1e-6, 1e-4) if precision == "f32" else (4e-3 * amax, 4e-4 * amax, 1e-2)
    assert err.max().item() <= tol_max and err.mean().item() <= tol_mean and le <= tol_logit
    g = golden("embedder_xlsr2b_4s.npz")                                   # the reference's own extract_features + TorchLogReg
    assert tuple(hid.shape[1:]) == tuple(g["shape"]) == (199, 1920)
    assert (hid[0, :8, :16].cpu() - torch.from_numpy(g["corner"])).abs().max().item() <= tol_max
    assert (hid[0].mean(0).cpu() - torch.from_numpy(g["pooled"])).abs().max().item() <= tol_max
    assert abs(logit.item() - float(g["logit"].reshape(-1)[0])) <= tol_logit
    assert abs(prob.item() - float(g["prob"].reshape(-1)[0])) <= tol_logit
    # batch invariance at this width (utterances shard): clip 0 of a 3-clip batch is bit-identical to the single clip
    w3 = torch.cat([w, syn.make_clips(2, 64000, seed=5)], 0)
    hid3, _, _ = emb.forward(w3.to(gpu_device))
    assert torch.equal(hid3[0], hid[0])


def test_xlsr2b_input_gradient_1s(gpu_device, xlsr2b_model):
    """The fp32-class gradient chain at the reference's width (head dim 120: the one-matrix-in-LDS form of the fp32-MFMA
    attention backward, K = 7680 dgrad GEMMs), 1 s clip, against fp32 autograd through the oracle."""
    from addvisor_hip.embedder_grad import EmbedderGrad
    cfg, sd, coef, icpt = xlsr2b_model
    w = syn.make_clips(1, 16000, seed=3)
    eg = EmbedderGrad(HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision="f32"))
    eg.forward(w.to(gpu_device))
    dx = eg.backward().cpu()
    with torch.enable_grad():
        ref = attribution_ref.input_gradient(w, sd, cfg, coef, icpt)
    rel = ((dx - ref).abs().max() / ref.abs().max()).item()
    cos = torch.nn.functional.cosine_similarity(dx.double().flatten(), ref.double().flatten(), dim=0).item()
    print(f"XLS-R-2B shape input gradient [f32]: max rel err {rel:.3e}, cosine {cos:.8f}")
    assert rel < 1e-4 and cos > 0.999999


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_ig_50_steps_large(gpu_device, large_model, precision):
    """captum_saliency.py:131-135 at config 5's model and step count: IntegratedGradients(n_steps=50, gausslegendre, zero
    baseline), 2 clips x 1 s, path-batched in chunks of 20 rows (10 steps), against the oracle (torch autograd through the
    CPU restatement; parity unpinned: captum absent).  Stated tolerance: fp32-class chain (the reference's fp32 autograd
    class) 1e-4 of max|attr| and cosine > 0.999999; fp16 chain 3e-2 / 0.999."""
    from addvisor_hip.attribution import HipAttribution
    cfg, sd, coef, icpt = large_model
    att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision=precision))
    assert att.precision == precision
    w = syn.make_clips(2, 16000, seed=12)
    ours = att.integrated_gradients(w.to(gpu_device), n_steps=50, internal_batch_size=20)
    with torch.enable_grad():
        ref = attribution_ref.integrated_gradients(w, sd, cfg, coef, icpt, n_steps=50)
    rel = ((ours.cpu() - ref).abs().max() / ref.abs().max()).item()
    cos = torch.nn.functional.cosine_similarity(ours.cpu().double().flatten(), ref.double().flatten(), dim=0).item()
    print(f"IG 50 steps, wav2vec2-large [{precision}]: max rel err {rel:.3e}, cosine {cos:.8f}")
    tol, cmin = (1e-4, 0.999999) if precision == "f32" else (3e-2, 0.999)
    assert bool(torch.isfinite(ours).all()) and rel < tol and cos > cmin
    whole = att.integrated_gradients(w.to(gpu_device), n_steps=50, internal_batch_size=100)    # chunking does not change the sum order per clip
    assert ((whole - ours).abs().max() / ref.abs().max()).item() < (1e-5 if precision == "f32" else 1e-3)


# ------------------------------------------------------------------------------------------ config 4
def test_run_addvisor_metrics_ragged_batches_4s(gpu_device, capsys):
    """The dataset loop of LMAC_metrics.py:117-172 with the drop-in modules: 11 clips of 4 s, batch_size 4 => batches of
    4, 4 and 3 (ragged tail), wav2vec2-base; the five printed means against the oracle run over the same clips in one batch."""
    os.environ["ADDVISOR_EMBEDDER"] = "base"
    runtime.reset()
    try:
        import LMAC_metrics
        LMAC_metrics.audio_processor.audio_length = 4
        clips = syn.make_clips(11, 64000, seed=93)

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return 11

            def __getitem__(self, i):
                return clips[i].to(gpu_device), f"clip{i}.wav"

        m = LMAC_metrics.run_addvisor_metrics("", "", batch_size=4, dataset=DS())
        printed = capsys.readouterr().out.strip().splitlines()
        assert [l.split(":")[0].strip() for l in printed] == ["faithfulness", "fidelity", "average drop", "average increase", "average gain"]
        cfg, sd = runtime.embedder_config_and_weights()
        clf = runtime.classifier()
        ref = lmac_ref.explain(clips, sd, cfg, clf.coef_, clf.intercept_, syn.unet_weights(), audio_length=4)
        r = lmac_ref.lmac_summary(ref["predictions"], ref["theta_out"], ref["masked_predictions"])
        print("HIP", m, "oracle", r)
        assert runtime.hip_embedder().precision == "f32"                     # the default mode: metric means to 1e-5 (percent scales: 1e-3)
        assert abs(m["faithfulness"] - r["faithfulness"]) < 1e-5 and abs(m["fidelity"] - r["fidelity"]) < 1e-6
        assert abs(m["AD"] - r["AD"]) < 1e-3 and abs(m["AI"] - r["AI"]) < 1e-3 and abs(m["AG"] - r["AG"]) < 1e-3
    finally:
        LMAC_metrics.audio_processor.audio_length = 5
        os.environ.pop("ADDVISOR_EMBEDDER", None)
        runtime.reset()


def _sharded_metrics_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    dist.init_process_group("gloo", rank=rank, world_size=world)       # one GPU on the test box: the exchange is rehearsed over gloo
    import LMAC_metrics
    LMAC_metrics.audio_processor.audio_length = 1
    clips = syn.make_clips(7, 16000, seed=95)
    dev = torch.device("cuda:0")

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 7

        def __getitem__(self, i):
            return clips[i].to(dev), f"clip{i}.wav"

    m = LMAC_metrics.run_addvisor_metrics("", "", batch_size=2, dataset=DS())
    q.put((rank, m))
    dist.destroy_process_group()


def test_run_addvisor_metrics_sharded_two_ranks(gpu_device, capsys):
    """BASELINE config 4's structure through the drop-in module: under torch.distributed run_addvisor_metrics walks each rank's
    contiguous block (7 clips -> 4 + 3), combines the per-clip probabilities with one all_gather into clip order and reduces
    them identically everywhere: both ranks return the SAME five numbers, equal to the single-process run bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=600) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0] == res[1]
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    runtime.reset()
    try:
        import LMAC_metrics
        LMAC_metrics.audio_processor.audio_length = 1
        clips = syn.make_clips(7, 16000, seed=95)

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return 7

            def __getitem__(self, i):
                return clips[i].to(gpu_device), f"clip{i}.wav"

        single = LMAC_metrics.run_addvisor_metrics("", "", batch_size=2, dataset=DS())
    finally:
        LMAC_metrics.audio_processor.audio_length = 5
        os.environ.pop("ADDVISOR_EMBEDDER", None)
        runtime.reset()
    assert single == res[0], (single, res[0])


# ------------------------------------------------------------------------------------------ the RCCL branch, once
def _rccl_world1_worker(port, q):
    """One rank, backend "nccl" (= RCCL on ROCm), on the one GPU of the box: the exchange step of the path -- the fixed-order
    all_gather of per-clip probabilities and the metric reduction on CUDA tensors -- through the same code `bench.py --gpus N`
    and the sharded `run_addvisor_metrics` execute on a node.  Not a scaling measurement."""
    import torch.distributed as dist
    from addvisor_hip import pipeline as P
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(5)
        local = torch.rand(37, 3, generator=g).to(dev)
        allp = P.gather_probabilities(local, 37)                          # dist.all_gather on CUDA tensors over RCCL
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                          # bench.py's max-over-ranks timing reduction
        dist.barrier()
        m = P.lmac_metrics(allp[:, 0].contiguous(), allp[:, 1].contiguous(), allp[:, 2].contiguous())
        q.put((dist.get_backend(), bool(torch.equal(allp, local)), float(t.item()), m))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_runs_the_exchange_step_world1(gpu_device):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(36500 + (os.getpid() % 2000), q))
    p.start()
    backend, same, tmax, m = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0 and backend == "nccl" and same and tmax == 1.25
    g = torch.Generator().manual_seed(5)
    local = torch.rand(37, 3, generator=g)
    ref = lmac_ref.lmac_summary(local[:, 0:1], local[:, 1:2], local[:, 2:3])
    for k in ref:
        assert abs(m[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (k, m[k], ref[k])
