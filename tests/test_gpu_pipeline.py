"""GPU parity of the whole explanation path and of the LMAC metric kernel against the CPU oracle."""
import numpy as np
import pytest
import torch

from addvisor_hip import pipeline as P, synthetic as syn
from oracle import lmac_ref

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

# Stated tolerances per precision mode.  "f32" = the default fp32-class mode (what a test without an explicit precision runs:
# the reference's arithmetic class); "f16" = fp16 GEMM operands (measured ~2e-3 on probabilities).
TOL = {"f32": dict(prob=1e-4, mask=2e-5, wave=5e-5), "f16": dict(prob=1e-2, mask=1.5e-2, wave=2e-2)}
TOL_PROB = TOL["f16"]["prob"]


def test_default_precision_is_f32(gpu_device, monkeypatch):
    monkeypatch.delenv("ADDVISOR_PRECISION", raising=False)
    assert P.default_precision() == "f32"


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("domain", ["log1p", "linear"])
def test_explain_tiny_vs_oracle(gpu_device, domain, precision):
    cfg = syn.tiny_config(False)
    emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1, domain=domain, precision=precision)
    w = syn.make_clips(4, 16000, seed=77)
    out = pipe.explain(w.to(gpu_device), keep=True)
    ref = lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=1, domain=domain)
    t = TOL[precision]
    for k in ("predictions", "theta_out", "masked_predictions"):
        assert (out[k].cpu() - ref[k]).abs().max().item() <= t["prob"], k
    assert (out["mask"].cpu() - ref["mask"]).abs().max().item() <= t["mask"]
    assert (out["wave_in"].cpu() - ref["wave_in"]).abs().max().item() <= t["wave"]      # follows the mask tolerance
    assert (out["wave_out"].cpu() - ref["wave_out"]).abs().max().item() <= t["wave"]
    if precision == "f32":
        assert torch.equal(out["mask"].cpu() > 0.5, ref["mask"] > 0.5)                  # mask indices: exact


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("seconds", [4, 5])
def test_explain_base_4s_vs_oracle(gpu_device, seconds, precision):
    """BASELINE models (wav2vec2-base + U-Net) on two 4 s clips, and on the reference's default ``audio_length=5``
    (T = 249 frames, 512 x 248 U-Net grid: SURVEY.md §8 sizes in brackets)."""
    cfg = syn.base_config()
    emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=seconds, precision=precision)
    w = syn.make_clips(2, 16000 * seconds)
    out = pipe.explain(w.to(gpu_device))
    ref = lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=seconds)
    for k in ("predictions", "theta_out", "masked_predictions"):
        err = (out[k].cpu() - ref[k]).abs().max().item()
        print(k, precision, "max err", err, "values", out[k].view(-1).tolist(), ref[k].view(-1).tolist())
        assert err <= TOL[precision]["prob"], k
    if precision == "f32":
        assert torch.equal(out["mask"].cpu() > 0.5, ref["mask"] > 0.5)


def test_lmac_metrics_kernel(gpu_device):
    r = np.random.Generator(np.random.PCG64(3))
    n = 1000
    p, t, o = (torch.from_numpy(r.uniform(0, 1, size=(n, 1)).astype(np.float32)) for _ in range(3))
    p[:5, 0] = torch.tensor([0.9, 0.2, 0.5, 0.5, 0.7])            # ties at 0.5 included
    t[:5, 0] = torch.tensor([0.8, 0.1, 0.6, 0.5, 0.3])
    o[:5, 0] = torch.tensor([0.4, 0.6, 0.5, 0.2, 0.9])
    d = gpu_device
    got, pc = P.lmac_metrics(p.to(d), t.to(d), o.to(d), per_clip=True)
    ref = lmac_ref.lmac_summary(p, t, o)
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (k, got[k], ref[k])
    pc = pc.cpu()
    assert torch.equal(pc[0], lmac_ref.compute_faithfulness(p, o))          # per-clip values: bit-exact fp32
    assert torch.equal(pc[1], lmac_ref.compute_fidelity(t, p).view(-1))
    assert torch.allclose(pc[2], lmac_ref.compute_AD(t, p), rtol=1e-6, atol=1e-6)
    assert torch.equal(pc[3], lmac_ref.compute_AI(t, p))
    assert torch.allclose(pc[4], lmac_ref.compute_AG(t, p), rtol=1e-6, atol=1e-5)


def test_lmac_metrics_kernel_golden(gpu_device, golden):
    """advh_lmac_metrics_accumulate against the reference's own metric functions (tests/golden/lmac_metrics.npz, generated
    from LMAC_metrics.py:31-73): per-clip faithfulness / fidelity / AI bit-exact, AD / AG to 1e-6 relative (one division),
    the five dataset means to 1e-6."""
    g = golden("lmac_metrics.npz")
    d = gpu_device
    p, t, o = (torch.from_numpy(g[k]).to(d) for k in ("predictions", "theta_out", "masked_predictions"))
    got, pc = P.lmac_metrics(p, t, o, per_clip=True)
    pc = pc.cpu()
    assert torch.equal(pc[0], torch.from_numpy(g["faithfulness"]))
    assert torch.equal(pc[1], torch.from_numpy(g["fidelity"]))
    assert torch.allclose(pc[2], torch.from_numpy(g["AD"]), rtol=1e-6, atol=1e-6)
    assert torch.equal(pc[3], torch.from_numpy(g["AI"]))
    assert torch.allclose(pc[4], torch.from_numpy(g["AG"]), rtol=1e-6, atol=1e-5)
    for k, v in zip(P.METRIC_NAMES, g["means"]):
        assert abs(got[k] - float(v)) <= 1e-6 * max(1.0, abs(float(v))), (k, got[k], v)


def test_full_batch_properties(gpu_device):
    """BASELINE batch (64 x 4 s): size-independent properties, no oracle needed."""
    cfg = syn.base_config()
    pipe = P.ExplainPipeline(cfg, syn.embedder_weights(cfg), *syn.logreg_weights(cfg.hidden_size), syn.unet_weights(),
                             gpu_device, audio_length=4, domain="linear")
    w = syn.make_clips(64, 64000).to(gpu_device)
    out = pipe.explain(w, keep=True)
    again = pipe.explain(w)
    for k in ("predictions", "theta_out", "masked_predictions", "mask"):
        assert torch.equal(out[k], again[k])                                 # run-to-run determinism
        assert torch.isfinite(out[k]).all()
    assert ((out["mask"] >= 0) & (out["mask"] <= 1)).all()
    # linear masking: mask-in + mask-out resynthesis reconstructs the clip
    assert (out["wave_in"] + out["wave_out"] - w).abs().max().item() < 2e-5
    # a sub-batch gives bit-identical per-clip results (utterance independence => sharding is exact)
    sub = pipe.explain(w[8:24].contiguous())
    assert torch.equal(sub["theta_out"], out["theta_out"][8:24])
    assert torch.equal(sub["predictions"], out["predictions"][8:24])


def test_explain_with_vocoder_resynthesis(gpu_device):
    """The "masked spectrogram -> HiFi-GAN vocoder -> classifier re-forward" variant: the mask-in / mask-out clips are
    re-rendered through the mel front end (hifigan.py:163-178) and the V1 generator before the classifier."""
    from addvisor_hip.hifigan import HipHifigan
    from oracle import hifigan_ref, signal_ref, wav2vec2_ref
    cfg, hcfg = syn.tiny_config(False), syn.hifigan_tiny_config()
    emb_sd, unet_sd, hsd = syn.embedder_weights(cfg), syn.unet_weights(), syn.hifigan_weights(hcfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    w = syn.make_clips(2, 16000, seed=78)
    plain = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1, precision="f16")
    base = plain.explain(w.to(gpu_device), keep=True)
    # the tiny generator takes 16 mel bands: use the first 16 rows of the 80-band mel on both sides
    class Voc16(HipHifigan):
        def decode_batch(self, mel):
            return super().decode_batch(mel[:, :16].contiguous())
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1, vocoder=Voc16(hcfg, hsd, gpu_device, precision="f16"),
                             precision="f16")
    out = pipe.explain(w.to(gpu_device), keep=True)
    assert torch.equal(out["predictions"], base["predictions"]) and torch.equal(out["mask"], base["mask"])
    for key, prob in (("wave_in", "theta_out"), ("wave_out", "masked_predictions")):
        pre = base[key].cpu()                                                        # the ISTFT resynthesis (parity: test above)
        voc = hifigan_ref.generator(signal_ref.mel_spectrogram(pre)[:, :16], hsd, hcfg)[:, 0, :16000]
        assert (out[key].cpu() - voc).abs().max().item() <= 2e-2                     # tests/test_gpu_hifigan.py tolerance
        _, p_ref = wav2vec2_ref.classify(voc, emb_sd, cfg, coef, icpt)
        assert (out[prob].cpu() - p_ref).abs().max().item() <= TOL_PROB


def test_hip_graph_replay_matches_eager(gpu_device):
    """explain() captured into a HIP graph and replayed on new inputs is bit-identical to the eager launch sequence."""
    cfg = syn.tiny_config(False)
    emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1)
    pipe.capture(3)
    for seed in (91, 92):
        w = syn.make_clips(3, 16000, seed=seed).to(gpu_device)
        eager = {k: v.clone() for k, v in pipe.explain(w).items()}
        replay = pipe.explain_graphed(w)
        torch.cuda.synchronize()
        for k in ("predictions", "theta_out", "masked_predictions", "mask"):
            assert torch.equal(eager[k], replay[k]), k
    short = syn.make_clips(3, 9000, seed=93).to(gpu_device)                 # ragged input: zero-padded to the clip length
    assert torch.equal(pipe.explain(short)["predictions"], pipe.explain_graphed(short)["predictions"])


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_explain_with_v1_vocoder_80_mels(gpu_device, precision):
    """north_star: "masked spectrogram -> HiFi-GAN vocoder -> classifier re-forward" with the real V1 generator on 80 mel
    bands (the tiny-generator test above covers the plumbing): 1 s clips, against the oracle run with the same vocoder step
    (oracle/lmac_ref.explain(vocoder=...)).  The vocoder runs at the PATH's precision (a generator handed over at another one
    is rebuilt).  Stated tolerances: f32 -- vocoded waveforms 1e-4, probabilities 1e-4; f16 -- 2e-2 / 1e-2."""
    from addvisor_hip.hifigan import HipHifigan
    cfg, hcfg = syn.tiny_config(False), syn.HifiganConfig()
    emb_sd, unet_sd, hsd = syn.embedder_weights(cfg), syn.unet_weights(), syn.hifigan_weights(hcfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    w = syn.make_clips(2, 16000, seed=79)
    other = "f16" if precision == "f32" else "f32"
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1, vocoder=HipHifigan(hcfg, hsd, gpu_device, precision=other),
                             precision=precision)
    assert pipe.vocoder.precision == precision == pipe.embedder.precision
    out = pipe.explain(w.to(gpu_device), keep=True)
    ref = lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=1, vocoder=(hsd, hcfg))
    tw, tp = (1e-4, 1e-4) if precision == "f32" else (2e-2, 1e-2)
    for key in ("wave_in", "wave_out"):
        err = (out[key].cpu() - ref[key]).abs().max().item()
        print(key, precision, "vocoded max err", err)
        assert err <= tw
    for k in ("predictions", "theta_out", "masked_predictions"):
        err = (out[k].cpu() - ref[k]).abs().max().item()
        print(k, precision, "max err", err)
        assert err <= tp, k


def test_explain_with_v1_vocoder_base_4s_f32(gpu_device):
    """The north-star variant at BASELINE sizes: wav2vec2-base, two 4 s clips, HiFi-GAN V1 (80 mels, 251 frames) in the loop,
    everything in the fp32-class mode, against oracle/lmac_ref.explain(vocoder=...): probabilities 1e-4, vocoded waveforms 1e-4,
    mask indices exact."""
    from addvisor_hip.hifigan import HipHifigan
    cfg, hcfg = syn.base_config(), syn.HifiganConfig()
    emb_sd, unet_sd, hsd = syn.embedder_weights(cfg), syn.unet_weights(), syn.hifigan_weights(hcfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    w = syn.make_clips(2, 64000, seed=80)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=4, vocoder=HipHifigan(hcfg, hsd, gpu_device), precision="f32")
    assert pipe.vocoder.precision == "f32"
    out = pipe.explain(w.to(gpu_device), keep=True)
    ref = lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=4, vocoder=(hsd, hcfg))
    assert torch.equal(out["mask"].cpu() > 0.5, ref["mask"] > 0.5)
    for key in ("wave_in", "wave_out"):
        err = (out[key].cpu() - ref[key]).abs().max().item()
        print(key, "vocoded max err", err)
        assert err <= 1e-4
    for k in ("predictions", "theta_out", "masked_predictions"):
        err = (out[k].cpu() - ref[k]).abs().max().item()
        print(k, "max err", err, out[k].view(-1).tolist(), ref[k].view(-1).tolist())
        assert err <= 1e-4, k


@pytest.mark.parametrize("B,n", [(1, 16000), (3, 11111), (2, 20001)])
def test_explain_edge_shapes_f32(gpu_device, B, n):
    """Ragged inputs of the loop body (LMAC_metrics.py:117-157 feeds whatever the loader returns): a single clip, clips
    shorter than audio_length (zero-padded tail, audioprocessor.py:83-98) and longer (cropped), odd sample counts; fp32-class
    mode against the oracle: probabilities 1e-4, mask 2e-5, mask indices equal, waveforms 5e-5."""
    cfg = syn.tiny_config(True)
    emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=1, precision="f32")
    w = syn.make_clips(B, n, seed=200 + n)
    out = pipe.explain(w.to(gpu_device), keep=True)
    ref = lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=1)
    for k in ("predictions", "theta_out", "masked_predictions"):
        assert out[k].shape == (B, 1) and (out[k].cpu() - ref[k]).abs().max().item() <= 1e-4, k
    assert (out["mask"].cpu() - ref["mask"]).abs().max().item() <= 2e-5
    assert torch.equal(out["mask"].cpu() > 0.5, ref["mask"] > 0.5)
    assert (out["wave_in"].cpu() - ref["wave_in"]).abs().max().item() <= 5e-5
    assert (out["wave_out"].cpu() - ref["wave_out"]).abs().max().item() <= 5e-5


def test_full_batch_f32_mask_indices_match_oracle(gpu_device):
    """BASELINE batch (64 x 4 s, wav2vec2-base + U-Net) in the fp32-class mode: for four clips spread over the batch the
    `mask > 0.5` index set equals the oracle's bit for bit and the three probabilities agree to 1e-4 (the oracle runs those
    four clips only; sub-batch bit-equality, tested above, covers the rest of the batch)."""
    cfg = syn.base_config()
    emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, gpu_device, audio_length=4, precision="f32")
    w = syn.make_clips(64, 64000)
    out = pipe.explain(w.to(gpu_device))
    pick = [0, 21, 42, 63]
    ref = lmac_ref.explain(w[pick], emb_sd, cfg, coef, icpt, unet_sd, audio_length=4)
    mask = out["mask"][pick].cpu()
    flips = int(((mask > 0.5) != (ref["mask"] > 0.5)).sum())
    print(f"full batch f32: mask max err {(mask - ref['mask']).abs().max():.3e}, index flips {flips} of {mask.numel()}, "
          f"closest reference value to 0.5: {(ref['mask'] - 0.5).abs().min():.2e}")
    assert flips == 0
    for k in ("predictions", "theta_out", "masked_predictions"):
        assert (out[k][pick].cpu() - ref[k]).abs().max().item() <= 1e-4, k
